"""The ``!fire`` command's arithmetic as pure functions.

The reference does this inline in its chat bot, between network and database
calls (``nerf-gun-control/main.py:421-441, 478-491, 494-609``); nothing there
tests it and the module cannot be imported without ``twitchio`` and
``aiomysql``.  PARITY UNPINNED: the expected values in this module's tests
were worked out by hand from the cited lines, not produced by the reference.

Per command the work is two integer adds, four compares, two multiplies, two
``max`` and two subtracts.  It is host logic; nothing here is a candidate for
a GPU.

Reference behaviour kept, although it is a defect:

* a negative shot count gets through.  The spaced spelling takes any ``int``
  (``main.py:485``) and only zero is refused (``:499``), so ``!fire 10 10 -2``
  parses to ``(10, 10, -2)``, :func:`aim` accepts it, and
  :func:`can_afford` prices it at a negative cost that any wallet covers.  The
  request then reaches the gun server with ``shot=-2``; what a real gun does
  with it is outside the reference.  These functions restate what the
  reference computes, so they keep it, and the tests state it.
  :func:`settle` is where it stops: a negative count *reported by the gun
  path* means "nothing was fired" in the reference (``:582``, ``:864-865``)
  and never touches the wallet.

Reference behaviour not reproduced:

* the ``NameError`` on every successful shot by anyone but the channel owner
  (``remaining_credits`` is read at ``main.py:600`` and assigned only on the
  owner branch at ``:595``);
* the chat text of the zero-shot refusal is misspelt in the reference
  (``"minumum"``, ``main.py:500``); the string here is the corrected one the
  reference itself writes to its OBS log on the next line (``:501``).
"""

import re
from dataclasses import dataclass

# reference main.py:430 -- "!!5,6,7", "!f5,6,7", "!fire5,6,7"; digits only, so
# the compact form cannot carry a negative angle or a negative shot count
_COMPACT = re.compile(r"^!(!|f|fire)(\d+),(\d+),(\d+)$")
# reference main.py:478 -- command "fire" with aliases "f" and "!", prefix "!"
_SPACED = re.compile(r"^!(fire|f|!)(?:\s+(.*))?$", re.S)

USAGE = "Usage: !fire 30 45 2  or  !f 30,45,2  or  !!30,45,2"

# What the bot back-fills into an empty ``system_config`` table (reference
# main.py:169-180).  Chat angles 45..225 / 45..135 land inside these limits.
DEFAULT_GUN_CONFIG = {
    "min_horizontal": 0,
    "max_horizontal": 180,
    "min_vertical": 0,
    "max_vertical": 90,
    "horizontal_offset": -45,
    "vertical_offset": -45,
}

# What a database built from the shipped schema holds (reference
# nerf-admin/nerfdb-setup.sql:40-43, 49-50), and what the bot falls back to
# when a row is missing or the query fails (limits from params.py:54-57,
# offsets from main.py:384, 394, 409-410).  ``ensure_system_config_defaults``
# only inserts missing rows (main.py:190-194), so on such a database this set
# is the one in force.  Chat angles 0..90 / 60..120 land inside these limits,
# and the post-offset angles are the gun-frame ranges the burst script draws
# from directly (fastshooter.py:10: x in [-45, 45], y in [0, 60]) -- those
# ranges fit this set, not DEFAULT_GUN_CONFIG.
SCHEMA_GUN_CONFIG = {
    "min_horizontal": -45,
    "max_horizontal": 45,
    "min_vertical": 0,
    "max_vertical": 60,
    "horizontal_offset": -45,
    "vertical_offset": -60,
}

# Returned by the reference's ``do_fire`` when the gun is switched off in the
# database (main.py:864-865); the caller then reports "Gun INACTIVE" and
# leaves the wallet alone (:582, :597, :606-609).
GUN_INACTIVE = -1


def parse_fire_command(content):
    """``(x, y, shots)`` from a chat line, ``None`` if it is not a fire command
    or its arguments do not parse (the bot then answers :data:`USAGE`).

    Two spellings (reference ``main.py:430-439`` and ``:478-491``): the compact
    ``!<alias><x>,<y>,<z>`` and the spaced ``!<alias> x y z`` where commas count
    as spaces and arguments past the third are ignored.  The spaced spelling
    accepts signed integers, the shot count included (``:485``).
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

    Reference ``main.py:499-521``: exactly zero shots is refused first (a
    negative count is not); the offsets are added before the bounds are
    checked, and the bounds are inclusive.  The refusal text follows
    ``main.py:501``.
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
    request (reference ``main.py:567-568``).  A negative ``shots`` gives a
    negative cost, which every wallet covers, as in the reference."""
    total_cost = credits_per_shot * shots
    return (current_credits + bonus_credits) >= total_cost, total_cost


def settle(current_credits, bonus_credits, credits_per_shot, shots_fired):
    """``(remaining_daily, remaining_bonus)`` after ``shots_fired`` darts left
    the gun: daily credits are spent first, the overflow comes out of the bonus
    (reference ``main.py:584-587``).  Charged on the shots the gun reports, not
    on the shots requested.

    A negative ``shots_fired`` leaves both balances as they were.  The
    reference settles only under ``if shots_fired >= 0`` (``main.py:582``):
    ``do_fire`` returns -1 for a gun that is switched off (``:864-865``), and
    the client's shot count is -1 when a ``/nerf`` reply carries a ``message``
    without a ``shots:`` tag (``nerf_controller.py:21``).  Without the guard
    the arithmetic below would add credits.
    """
    if shots_fired < 0:
        return current_credits, bonus_credits
    used = shots_fired * credits_per_shot
    remaining_daily = max(0, current_credits - used)
    from_bonus = max(0, used - current_credits)
    return remaining_daily, bonus_credits - from_bonus
