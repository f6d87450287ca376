"""The sequential burst (reference nerf-gun-control/fastshooter.py:10-19)."""

import random

from nerf_amd import NerfController
from nerf_amd.fastshooter import X_RANGE, Y_RANGE, burst, main, targets
from nerf_amd.simulator import KO


def test_targets_are_in_the_reference_ranges_and_seeded():
    coords = targets(100, random.Random(7))
    assert len(coords) == 100
    assert all(X_RANGE[0] <= x <= X_RANGE[1] and Y_RANGE[0] <= y <= Y_RANGE[1] for x, y in coords)
    assert coords == targets(100, random.Random(7))


def test_burst_fires_every_target_in_order(simulator):
    seen = []
    coords = targets(5, random.Random(1))
    fired, failure = burst(NerfController(simulator.url), coords,
                           on_shot=lambda x, y, status: seen.append((x, y, status["shots"])))
    assert (fired, failure) == (5, None)
    assert seen == [(x, y, 1) for x, y in coords]


def test_burst_stops_at_the_first_failure(simulator):
    calls = []

    def break_gun_after_two(x, y, status):
        calls.append((x, y))
        if len(calls) == 2:
            simulator.gun.force(KO)

    fired, failure = burst(NerfController(simulator.url), targets(10, random.Random(2)),
                           on_shot=break_gun_after_two)
    assert fired == 2 and len(calls) == 2
    assert failure["status"] == "error" and "503" in failure["message"]


def test_empty_burst(simulator):
    assert burst(NerfController(simulator.url), []) == (0, None)


def test_command_line_fires_the_seeded_burst(simulator, capsys):
    assert main([simulator.url, "-n", "2", "--seed", "1"]) == 0
    expected = targets(2, random.Random(1))
    assert capsys.readouterr().out.splitlines() == [f"Fired at x={x}, y={y}" for x, y in expected]


def test_command_line_reports_the_first_failure(simulator, capsys):
    simulator.gun.force(KO)
    assert main([simulator.url, "-n", "3", "--seed", "1"]) == 1
    out = capsys.readouterr().out
    assert out.startswith("Error after 0 shots: ") and "503" in out
