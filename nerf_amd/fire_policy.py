"""The ``!fire`` command's arithmetic as pure functions.

The reference does this inline in its chat bot, between network and database
calls (``nerf-gun-control/main.py:421-441, 478-491, 494-609``); nothing there
tests it and the module cannot be imported without ``twitchio`` and
``aiomysql``.  PARITY UNPINNED: the expected values in this module's tests
were worked out by hand from the cited lines, not produced by the reference.

Per command the work is two integer adds, four compares, two multiplies, two
``max`` and two subtracts.  It is host logic; nothing here is a candidate for
a GPU.
"""

import re
from dataclasses import dataclass

# reference main.py:430 -- "!!5,6,7", "!f5,6,7", "!fire5,6,7"; digits only, so
# the compact form cannot carry a negative angle
_COMPACT = re.compile(r"^!(!|f|fire)(\d+),(\d+),(\d+)$")
# reference main.py:478 -- command "fire" with aliases "f" and "!", prefix "!"
_SPACED = re.compile(r"^!(fire|f|!)(?:\s+(.*))?$", re.S)

USAGE = "Usage: !fire 30 45 2  or  !f 30,45,2  or  !!30,45,2"

# reference main.py:169-180
DEFAULT_GUN_CONFIG = {
    "min_horizontal": 0,
    "max_horizontal": 180,
    "min_vertical": 0,
    "max_vertical": 90,
    "horizontal_offset": -45,
    "vertical_offset": -45,
}


def parse_fire_command(content):
    """``(x, y, shots)`` from a chat line, ``None`` if it is not a fire command
    or its arguments do not parse (the bot then answers :data:`USAGE`).

    Two spellings (reference ``main.py:430-439`` and ``:478-491``): the compact
    ``!<alias><x>,<y>,<z>`` and the spaced ``!<alias> x y z`` where commas count
    as spaces and arguments past the third are ignored.
    """
    text = content.strip()
    m = _COMPACT.match(text)
    if m:
        return int(m.group(2)), int(m.group(3)), int(m.group(4))
    m = _SPACED.match(text)
    if not m:
        return None
    args = (m.group(2) or "").replace(",", " ").split()
    if len(args) < 3:
        return None
    try:
        return int(args[0]), int(args[1]), int(args[2])
    except ValueError:
        return None


@dataclass(frozen=True)
class Aim:
    ok: bool
    x: int
    y: int
    error: str = ""


def aim(x, y, shots, config=DEFAULT_GUN_CONFIG):
    """Apply the mounting offsets and check the angle limits.

    Reference ``main.py:499-521``: zero shots is refused first; the offsets
    are added before the bounds are checked, and the bounds are inclusive.
    """
    if shots == 0:
        return Aim(False, x, y, "Can't shoot 0 shots, minimum is 1!")
    ax = x + config.get("horizontal_offset", 0)
    ay = y + config.get("vertical_offset", 0)
    lo_h, hi_h = config["min_horizontal"], config["max_horizontal"]
    lo_v, hi_v = config["min_vertical"], config["max_vertical"]
    if ax < lo_h or ax > hi_h or ay < lo_v or ay > hi_v:
        return Aim(False, ax, ay,
                   f"Fire command out of bounds. Horizontal: {lo_h} to {hi_h}, "
                   f"Vertical: {lo_v} to {hi_v}")
    return Aim(True, ax, ay)


def can_afford(current_credits, bonus_credits, credits_per_shot, shots):
    """``(ok, total_cost)``: daily plus bonus credits must cover the whole
    request (reference ``main.py:567-568``)."""
    total_cost = credits_per_shot * shots
    return (current_credits + bonus_credits) >= total_cost, total_cost


def settle(current_credits, bonus_credits, credits_per_shot, shots_fired):
    """``(remaining_daily, remaining_bonus)`` after ``shots_fired`` darts left
    the gun: daily credits are spent first, the overflow comes out of the bonus
    (reference ``main.py:584-587``).  Charged on the shots the gun reports, not
    on the shots requested."""
    used = shots_fired * credits_per_shot
    remaining_daily = max(0, current_credits - used)
    from_bonus = max(0, used - current_credits)
    return remaining_daily, bonus_credits - from_bonus
