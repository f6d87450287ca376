"""Fire-command arithmetic.  PARITY UNPINNED: the reference has no test for
main.py and the module cannot be imported here; every expected value below is
hand-derived from the cited reference lines."""

import pytest

from nerf_amd.fire_policy import (
    DEFAULT_GUN_CONFIG,
    aim,
    can_afford,
    parse_fire_command,
    settle,
)


@pytest.mark.parametrize(
    "line, expected",
    [
        ("!fire 30 45 2", (30, 45, 2)),  # main.py:478-486
        ("!f 30,45,2", (30, 45, 2)),  # commas count as blanks, :481
        ("!! 30, 45 ,2", (30, 45, 2)),
        ("!fire -10 5 1", (-10, 5, 1)),  # spaced form takes signs
        ("!fire 1 2 3 4 5", (1, 2, 3)),  # extras ignored, :483-485
        ("  !fire 1 2 3  ", (1, 2, 3)),  # stripped, :427
        ("!!5,6,7", (5, 6, 7)),  # compact forms, :430
        ("!f5,6,7", (5, 6, 7)),
        ("!fire5,6,7", (5, 6, 7)),
        ("!fire 1 2", None),  # too few -> usage, :491
        ("!fire a b c", None),
        ("!fire 1.5 2 3", None),
        ("!fire", None),
        ("!f-5,6,7", None),  # compact form is digits only
        ("!fired 1 2 3", None),
        ("fire 1 2 3", None),
        ("hello", None),
        ("", None),
    ],
)
def test_parse_fire_command(line, expected):
    assert parse_fire_command(line) == expected


def test_aim_applies_offsets_then_inclusive_bounds():
    # defaults main.py:169-180: offsets -45/-45, limits 0..180 / 0..90
    a = aim(90, 90, 1)
    assert (a.ok, a.x, a.y) == (True, 45, 45)
    assert aim(45, 45, 1).ok  # lands on (0, 0): inclusive lower bounds, :513-516
    assert aim(225, 135, 1).ok  # lands on (180, 90): inclusive upper bounds
    assert not aim(44, 45, 1).ok
    assert not aim(45, 44, 1).ok
    assert not aim(226, 135, 1).ok
    assert not aim(225, 136, 1).ok


def test_aim_out_of_bounds_message_names_the_limits():
    # main.py:518
    a = aim(0, 0, 1)
    assert a.error == "Fire command out of bounds. Horizontal: 0 to 180, Vertical: 0 to 90"


def test_aim_refuses_zero_shots_before_anything_else():
    # main.py:499-502 comes before the bounds check
    a = aim(10_000, 10_000, 0)
    assert not a.ok and "0 shots" in a.error


def test_aim_without_offsets_in_config():
    cfg = {k: v for k, v in DEFAULT_GUN_CONFIG.items() if not k.endswith("_offset")}
    a = aim(10, 20, 1, cfg)  # .get(..., 0) at main.py:508-509
    assert (a.ok, a.x, a.y) == (True, 10, 20)


@pytest.mark.parametrize(
    "current, bonus, cps, shots, ok, cost",
    [
        (10, 0, 1, 10, True, 10),  # exactly enough, main.py:567-568
        (10, 0, 1, 11, False, 11),
        (3, 4, 2, 3, True, 6),  # bonus counts
        (3, 2, 2, 3, False, 6),
        (0, 0, 1, 1, False, 1),
        (0, 0, 0, 5, True, 0),  # free tier
    ],
)
def test_can_afford(current, bonus, cps, shots, ok, cost):
    assert can_afford(current, bonus, cps, shots) == (ok, cost)


@pytest.mark.parametrize(
    "current, bonus, cps, fired, daily, bonus_left",
    [
        (10, 5, 1, 3, 7, 5),  # daily covers it, main.py:584-587
        (10, 5, 1, 10, 0, 5),  # daily exactly spent
        (10, 5, 1, 12, 0, 3),  # overflow from bonus
        (0, 5, 2, 2, 0, 1),
        (10, 5, 3, 0, 10, 5),  # gun reported no shots: nothing charged
        (4, 0, 1, 2, 2, 0),
    ],
)
def test_settle(current, bonus, cps, fired, daily, bonus_left):
    assert settle(current, bonus, cps, fired) == (daily, bonus_left)


def test_settle_conserves_credits_when_affordable():
    for current in range(0, 6):
        for bonus in range(0, 6):
            for cps in range(0, 3):
                for fired in range(0, 6):
                    if not can_afford(current, bonus, cps, fired)[0]:
                        continue
                    daily, bonus_left = settle(current, bonus, cps, fired)
                    assert daily >= 0 and bonus_left >= 0
                    assert daily + bonus_left == current + bonus - cps * fired
