"""Sequential burst of single shots at random angles.

The reference's script (``nerf-gun-control/fastshooter.py:10-19``) draws 100
``(x in [-45, 45], y in [0, 60])`` pairs and fires one dart at each, waiting
for the gun to go idle before the next and stopping at the first failure.  The
gun is one serial device, so the burst is sequential by construction; this is
the same loop as a function, with the random source and the count as
arguments.

Run as ``python -m nerf_amd.fastshooter URL [-n COUNT]``.
"""

import argparse
import random

from .nerf_controller import NerfController

X_RANGE = (-45, 45)
Y_RANGE = (0, 60)


def targets(count=100, rng=None):
    rng = rng if rng is not None else random.Random()
    return [(rng.randint(*X_RANGE), rng.randint(*Y_RANGE)) for _ in range(count)]


def burst(controller, coords, on_shot=None):
    """Fire one dart at each of ``coords`` in order.

    Returns ``(fired, failure)``: how many fires completed, and the status of
    the one that failed (``None`` if none did; the burst ends there).
    """
    fired = 0
    for x, y in coords:
        ok, status = controller.fire(x, y, 1, wait=True)
        if not ok:
            return fired, status
        fired += 1
        if on_shot is not None:
            on_shot(x, y, status)
    return fired, None


def main(argv=None):
    parser = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    parser.add_argument("url", help="gun server base URL, e.g. http://localhost:5555")
    parser.add_argument("-n", "--count", type=int, default=100)
    parser.add_argument("--seed", type=int, default=None)
    args = parser.parse_args(argv)
    rng = random.Random(args.seed)
    fired, failure = burst(NerfController(args.url), targets(args.count, rng),
                           on_shot=lambda x, y, s: print(f"Fired at x={x}, y={y}"))
    if failure is not None:
        print(f"Error after {fired} shots: {failure}")
        return 1
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
