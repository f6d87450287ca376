"""HTTP client for the gun server.

Mirrors the public surface of the reference client
(``nerf-gun-control/nerf_controller.py:7-69``): a class ``NerfController`` with
``fire``, ``stop``, ``get_status`` and ``wait_until_idle``, the same argument
names and defaults, and the same error convention -- no method lets a transport
or parsing failure escape; each folds it into its return value.

Wire contract (what the gun server must speak):

* ``GET {url}/nerf?x=<int>&y=<int>&shot=<int>`` -> 2xx JSON object whose
  ``message`` string carries ``shots:<N>`` (reference ``:12-21``);
* ``GET {url}/status`` -> 2xx JSON ``{"status": "idle"|"busy"|"ko"[, "shots": N]}``
  (reference ``:40-47``, ``:55-67``);
* ``GET {url}/stop`` -> 2xx, body returned as text (reference ``:31-36``).

Deliberate differences from the reference, none visible through its tests:

* progress goes to the ``nerf_amd.nerf_controller`` logger, not to ``print``;
* the reference module's first line imports a name from ``tkinter`` that it
  never uses (``:1``), which makes it unimportable on a headless Python; there
  is no such import here;
* ``wait_until_idle`` with a timeout that has already expired before the first
  poll returns ``(False, {"status": "error", ...})``; the reference reaches its
  final ``return`` with ``status`` unbound (``:53-54, 69``) and raises;
* ``wait_until_idle`` called directly does not raise on a malformed status
  reply either.  A JSON object without a ``status`` key counts as "not idle
  yet" and is polled again like any unknown status word; a reply that is not
  a JSON object ends the wait with an error status.  The reference indexes
  ``status["status"]`` (``:56``) and raises ``KeyError`` / ``TypeError``;
* ``wait_until_idle`` measures its timeout on the monotonic clock; the
  reference uses the wall clock (``:53-54``), so a clock step during the wait
  shortens or stretches it there;
* an optional ``request_timeout`` (seconds) bounds each HTTP request.  The
  reference passes none (``:15, 34, 44``), so one connection that never
  answers blocks it for good and the 45 s of ``wait_until_idle`` are never
  looked at.  The default stays ``None``, which keeps the call shape the
  reference's tests pin (no ``timeout`` keyword).
"""

import logging
import time

import requests

log = logging.getLogger(__name__)

_SHOTS_TAG = "shots:"


def _shots_from_message(message):
    """Number after the ``shots:`` tag of a ``/nerf`` reply, or -1 without a tag.

    Follows reference ``nerf_controller.py:19-21``: the text between the first
    and the second tag (or the end) must be an integer literal, otherwise
    ``int`` raises and ``fire`` reports the error.
    """
    pieces = message.split(_SHOTS_TAG)
    if len(pieces) < 2:
        return -1
    return int(pieces[1])


class NerfController:
    """Client for one gun server; holds its base URL and, optionally, a
    per-request timeout in seconds."""

    def __init__(self, server_url, request_timeout=None):
        # reference nerf_controller.py:9 -- tolerate "http://host:port/"
        self.server_url = server_url.rstrip("/")
        self.request_timeout = request_timeout

    def _get(self, path, params=None):
        url = self.server_url + path
        # The reference's tests pin the exact call shape: no `params` keyword
        # at all for /stop and /status, and never a `timeout` keyword
        # (test_nerf_controller.py:26, 46, 57) -- pass only what is set.
        extra = {}
        if params is not None:
            extra["params"] = params
        if self.request_timeout is not None:
            extra["timeout"] = self.request_timeout
        response = requests.get(url, **extra)
        response.raise_for_status()
        return response

    def fire(self, x=0, y=0, shot=1, wait=True):
        """Aim at (x, y) and fire ``shot`` darts.

        Returns ``(ok, info)``.  With ``wait`` true, ``info`` is the final
        status from :meth:`wait_until_idle`, which is told the shot count the
        server announced; otherwise ``info`` is the server's JSON reply.  Any
        failure gives ``(False, {"status": "error", "message": str(exc),
        "shots": 0})`` (reference ``nerf_controller.py:11-29``).
        """
        try:
            reply = self._get("/nerf", {"x": x, "y": y, "shot": shot}).json()
            log.debug("gun reply: %s", reply)
            shots = _shots_from_message(reply.get("message", _SHOTS_TAG + "0"))
            if not wait:
                return True, reply
            return self.wait_until_idle(shots=shots)
        except Exception as exc:  # never raise: reference :27-29
            log.warning("gun error: %s", exc)
            return False, {"status": "error", "message": str(exc), "shots": 0}

    def stop(self):
        """Ask the server to stop; its reply text, or ``"Error: ..."``
        (reference ``nerf_controller.py:31-38``)."""
        try:
            return self._get("/stop").text
        except Exception as exc:
            return f"Error: {exc}"

    def get_status(self):
        """The server's status JSON, or ``{"status": "error", "message": ...}``
        (reference ``nerf_controller.py:40-50``)."""
        try:
            status = self._get("/status").json()
            log.debug("gun status: %s", status)
            return status
        except Exception as exc:
            log.warning("status error: %s", exc)
            return {"status": "error", "message": str(exc)}

    def wait_until_idle(self, timeout=45, check_interval=0.05, shots=0):
        """Poll ``/status`` until the gun leaves ``busy`` or ``timeout`` seconds pass.

        Returns ``(True, status)`` on ``idle``, ``(False, status)`` on ``ko``,
        on ``error`` and on timeout (then with the last status seen).  A
        terminal status without a ``shots`` field gets one: the caller's
        ``shots`` for ``idle``, 0 for ``ko`` and ``error`` (reference
        ``nerf_controller.py:52-69``).
        """
        deadline = time.monotonic() + timeout
        status = {"status": "error", "message": "timeout expired before the first poll", "shots": 0}
        while time.monotonic() < deadline:
            status = self.get_status()
            if not isinstance(status, dict):
                return False, {"status": "error", "shots": 0,
                               "message": f"status reply is not a JSON object: {status!r}"}
            state = status.get("status")
            if state == "idle":
                status.setdefault("shots", shots)
                return True, status
            if state in ("ko", "error"):
                status.setdefault("shots", 0)
                return False, status
            time.sleep(check_interval)
        return False, status
