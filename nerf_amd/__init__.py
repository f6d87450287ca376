"""CPU-only restatement of the gun-control contract of Vadiml1024/nerf.

The reference is a Twitch chat bot that aims and fires a physical Nerf dart
gun over HTTP.  It has no numeric kernel, no tensor and no FFI seam, so this
package holds NO device code: no HIP sources, no compiled extension, nothing
that touches a GPU.  What it does hold is the one replaceable seam the
reference has, the HTTP contract between the bot and the gun server:

* :class:`NerfController` -- the client (reference
  ``nerf-gun-control/nerf_controller.py:7-69``), same method names, argument
  meaning and never-raise error convention;
* :class:`GunSimulator` -- a gun server that answers ``/nerf`` with the JSON
  the client parses (the reference's simulator answers in plain text, which the
  client cannot read: ``nerf-gun-simluator/simulator.py:62-66`` against
  ``nerf_controller.py:17-21``);
* :mod:`nerf_amd.fire_policy` -- the integer arithmetic of the ``!fire``
  command (reference ``nerf-gun-control/main.py:430-439, 478-491, 499-521,
  567-587``) as pure functions;
* :mod:`nerf_amd.fastshooter` -- the sequential burst script (reference
  ``nerf-gun-control/fastshooter.py:10-19``) as a callable.
"""

__all__ = ["NerfController", "GunSimulator"]

_HOME = {"NerfController": "nerf_controller", "GunSimulator": "simulator"}


def __getattr__(name):
    # Resolved on first use, so `python -m nerf_amd.simulator` does not find
    # its own module imported ahead of it by the package.
    if name in _HOME:
        from importlib import import_module

        return getattr(import_module(f"{__name__}.{_HOME[name]}"), name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
