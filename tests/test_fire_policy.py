"""Fire-command arithmetic.  PARITY UNPINNED: the reference has no test for
main.py and the module cannot be imported here; every expected value below is
hand-derived from the cited reference lines."""

import pytest

from nerf_amd.fire_policy import (
    DEFAULT_GUN_CONFIG,
    GUN_INACTIVE,
    SCHEMA_GUN_CONFIG,
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
        ("!fire 10 10 -2", (10, 10, -2)),  # the shot count too: plain int() at :485, only 0 refused at :499
        ("!f 10,10,-2", (10, 10, -2)),
        ("!!10,10,-2", None),  # compact form is digits only (:430); no blank, so no command "!" for :478 either
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


def test_aim_inclusive_corners_of_the_bot_defaults():
    # main.py:169-180 -- chat angles 45..225 / 45..135 map onto 0..180 / 0..90
    for x, y, gx, gy in [(45, 45, 0, 0), (225, 45, 180, 0), (45, 135, 0, 90), (225, 135, 180, 90)]:
        a = aim(x, y, 1, DEFAULT_GUN_CONFIG)
        assert (a.ok, a.x, a.y) == (True, gx, gy)
        for dx, dy in [(-1, 0), (1, 0), (0, -1), (0, 1)]:
            inside = 45 <= x + dx <= 225 and 45 <= y + dy <= 135
            assert aim(x + dx, y + dy, 1, DEFAULT_GUN_CONFIG).ok is inside


def test_aim_inclusive_corners_of_the_schema_defaults():
    # nerfdb-setup.sql:40-43, 49-50 (and params.py:54-57 with main.py:384, 394):
    # limits -45..45 / 0..60, offsets -45 / -60 -- chat angles 0..90 / 60..120
    assert SCHEMA_GUN_CONFIG == {
        "min_horizontal": -45, "max_horizontal": 45, "min_vertical": 0, "max_vertical": 60,
        "horizontal_offset": -45, "vertical_offset": -60,
    }
    for x, y, gx, gy in [(0, 60, -45, 0), (90, 60, 45, 0), (0, 120, -45, 60), (90, 120, 45, 60)]:
        a = aim(x, y, 1, SCHEMA_GUN_CONFIG)
        assert (a.ok, a.x, a.y) == (True, gx, gy)
        for dx, dy in [(-1, 0), (1, 0), (0, -1), (0, 1)]:
            inside = 0 <= x + dx <= 90 and 60 <= y + dy <= 120
            assert aim(x + dx, y + dy, 1, SCHEMA_GUN_CONFIG).ok is inside
    a = aim(0, 0, 1, SCHEMA_GUN_CONFIG)
    assert a.error == "Fire command out of bounds. Horizontal: -45 to 45, Vertical: 0 to 60"


def test_the_two_default_sets_disagree():
    # a command that is fine on a database the bot seeded itself is out of
    # bounds on one built from the shipped schema, and the other way round
    assert aim(200, 100, 1, DEFAULT_GUN_CONFIG).ok and not aim(200, 100, 1, SCHEMA_GUN_CONFIG).ok
    assert aim(10, 70, 1, SCHEMA_GUN_CONFIG).ok and not aim(10, 70, 1, DEFAULT_GUN_CONFIG).ok


def test_burst_script_ranges_are_gun_angles_of_the_schema_set():
    # fastshooter.py:10 draws x in [-45, 45], y in [0, 60] and sends them
    # straight to the gun: exactly the post-offset limits of the schema set
    from nerf_amd.fastshooter import X_RANGE, Y_RANGE

    assert X_RANGE == (SCHEMA_GUN_CONFIG["min_horizontal"], SCHEMA_GUN_CONFIG["max_horizontal"])
    assert Y_RANGE == (SCHEMA_GUN_CONFIG["min_vertical"], SCHEMA_GUN_CONFIG["max_vertical"])
    assert X_RANGE[0] < DEFAULT_GUN_CONFIG["min_horizontal"]  # would be out of bounds there


def test_aim_out_of_bounds_message_names_the_limits():
    # main.py:518
    a = aim(0, 0, 1)
    assert a.error == "Fire command out of bounds. Horizontal: 0 to 180, Vertical: 0 to 90"


def test_aim_refuses_zero_shots_before_anything_else():
    # main.py:499-502 comes before the bounds check
    a = aim(10_000, 10_000, 0)
    assert not a.ok and "0 shots" in a.error


def test_zero_shot_refusal_uses_the_corrected_spelling():
    # main.py:501 (OBS log), not the misspelt chat text at :500
    assert aim(90, 90, 0).error == "Can't shoot 0 shots, minimum is 1!"


def test_negative_shot_count_is_kept_as_the_reference_computes_it():
    # main.py:499 refuses only z == 0; :567-568 then prices -2 shots at -2,
    # which an empty wallet covers.  Kept on purpose, see the module docstring.
    a = aim(90, 90, -2)
    assert (a.ok, a.x, a.y) == (True, 45, 45)
    assert can_afford(0, 0, 1, -2) == (True, -2)
    assert can_afford(0, 0, 10, -1) == (True, -10)


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


@pytest.mark.parametrize("fired", [GUN_INACTIVE, -1, -2, -100])
def test_settle_leaves_the_wallet_alone_on_a_negative_count(fired):
    # main.py:582 -- `if shots_fired >= 0`; -1 is do_fire's "gun inactive"
    # (:864-865) and the client's "no shots: tag" (nerf_controller.py:21)
    assert settle(10, 5, 1, fired) == (10, 5)
    assert settle(0, 0, 3, fired) == (0, 0)
    assert settle(2, 9, 0, fired) == (2, 9)


def test_settle_never_adds_credits():
    for current in range(0, 5):
        for bonus in range(0, 5):
            for cps in range(0, 4):
                for fired in range(-3, 8):
                    daily, bonus_left = settle(current, bonus, cps, fired)
                    assert daily <= current and bonus_left <= bonus


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
