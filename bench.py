#!/usr/bin/env python3
"""Bench contract for a reference that has nothing to bench on a GPU.

Vadiml1024/nerf is a chat bot that fires a physical dart gun over HTTP; its
headline metric is recorded as "N/A: no data-parallel path".  This script
keeps the driver's contract -- ``python bench.py --gpus N --steps K --warmup W``
prints ONE JSON line from rank 0 -- and reports that honestly:

* ``value`` is null and ``roofline`` is null: there is no kernel, so there is
  no throughput on an MI355X and no bandwidth or MFMA fraction to state;
* no GPU is touched and torch is not imported, whatever ``--gpus`` says; under
  ``torch.distributed.run`` every rank but 0 returns at once (one gun is one
  serial device: the work neither shards nor replicates);
* the only thing that can be timed is the CPU workload the baseline names
  second: replayed ``!fire x y z`` chat lines, each parsed, bounds-checked,
  sent to the gun simulator over loopback HTTP, waited on until the gun is
  idle again, and settled against a credit balance.  One step is one such
  command.  It is reported under ``cpu_baseline``; with the reference's own
  actuation constant (0.5 s per shot) the time the gun stays busy is over
  90 % of every step and the client's 50 ms status-poll sleep most of the
  rest, which is the point: the path is bound by a physical actuator.
"""

import argparse
import json
import os
import random
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

from nerf_amd import GunSimulator, NerfController  # noqa: E402
from nerf_amd import fire_policy  # noqa: E402


def chat_lines(count, rng):
    """Synthetic chat: single-shot fire commands that land inside the default
    limits after the -45/-45 mounting offsets, in the three accepted spellings."""
    spell = ("!fire {x} {y} 1", "!f {x},{y},1", "!!{x},{y},1")
    return [
        spell[i % 3].format(x=rng.randint(45, 225), y=rng.randint(45, 135)) for i in range(count)
    ]


def handle(line, controller, wallet):
    """One chat command end to end; returns the shots the gun reported."""
    x, y, shots = fire_policy.parse_fire_command(line)
    target = fire_policy.aim(x, y, shots)
    if not target.ok:
        raise RuntimeError(target.error)
    affordable, _ = fire_policy.can_afford(wallet["daily"], wallet["bonus"], 1, shots)
    if not affordable:
        raise RuntimeError("synthetic wallet ran dry")
    ok, status = controller.fire(target.x, target.y, shots, wait=True)
    if not ok or status["shots"] < 0:
        # a negative count means nothing was fired (reference main.py:582, :864-865)
        raise RuntimeError(f"simulated gun failed: {status}")
    wallet["daily"], wallet["bonus"] = fire_policy.settle(
        wallet["daily"], wallet["bonus"], 1, status["shots"]
    )
    return status["shots"]


def metric_name():
    try:
        with open(os.path.join(HERE, "BASELINE.json")) as fh:
            return json.load(fh)["metric"]
    except (OSError, KeyError, ValueError):
        return "N/A: no data-parallel path"


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    ap.add_argument("--gpus", type=int, default=1, help="accepted for the contract; no GPU is used")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--seconds-per-shot", type=float, default=0.5,
                    help="simulated actuation time (reference simulator.py:73)")
    args = ap.parse_args(argv)

    if int(os.environ.get("RANK", "0")) != 0:
        return 0  # nothing to shard, nothing to replicate

    rng = random.Random(0)
    lines = chat_lines(args.warmup + args.steps, rng)
    wallet = {"daily": 10 ** 9, "bonus": 0}
    with GunSimulator(port=0, seconds_per_shot=args.seconds_per_shot, ko_probability=0.0) as sim:
        controller = NerfController(sim.url)
        for line in lines[: args.warmup]:
            handle(line, controller, wallet)
        t0 = time.perf_counter()
        shots = sum(handle(line, controller, wallet) for line in lines[args.warmup:])
        elapsed = time.perf_counter() - t0

    steps = max(args.steps, 1)
    ms_per_step = 1e3 * elapsed / steps
    out = {
        "metric": metric_name(),
        "value": None,
        "unit": None,
        "n_gpus": args.gpus,
        "gpus_used": 0,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True,
        "scaling": None,
        "vs_baseline": None,
        "dtype": "int",
        "data": "synthetic",
        "config": {
            "workload": "replayed '!fire x y 1' chat commands -> fire policy -> NerfController "
                        "-> gun simulator over loopback HTTP (BASELINE.json configs[1]); CPU only",
            "seconds_per_shot": args.seconds_per_shot,
            "parallelism": "none: one gun is one serial device",
        },
        "roofline": None,
        "roofline_note": "no kernel exists: the reference has no numeric hot path, so neither an "
                         "HBM nor an MFMA bound applies and no fraction is reported",
        "cpu_baseline": {
            "value": round(args.steps / elapsed, 3) if elapsed > 0 else None,
            "unit": "fire commands/s",
            "cores": 1,
            "kind": "port",
            "sample": f"{args.steps} sequential single-shot commands, {shots} shots, "
                      f"{args.seconds_per_shot} s simulated actuation per shot; "
                      f"the gun's busy time is {100 * args.seconds_per_shot * shots / elapsed:.1f}% "
                      f"of the timed region" if elapsed > 0 else "empty",
        },
    }
    print(json.dumps(out))
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
