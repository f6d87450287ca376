"""Gun server stand-in that speaks the contract :class:`NerfController` expects.

Same routes, status words and status codes as the reference's simulator
(``nerf-gun-simluator/simulator.py:22-90``):

* ``GET /nerf?x&y&shot`` and form-encoded ``POST /nerf`` start an actuation
  that keeps the gun ``busy`` for ``seconds_per_shot * shot`` seconds
  (reference ``:47-74``; 0.5 s per shot at ``:73``); 429 while busy, 503 while
  ``ko`` (``:49-54``);
* ``GET /status`` -> JSON ``{"status": "idle"|"busy"|"ko"}`` (``:85-90``);
* ``GET /stop`` -> text ``Nerf stopped``, gun idle (``:76-83``) -- a ``ko``
  gun included, as in the reference;
* any other path -> 404 (``:33-34``, ``:44-45``);
* every request first gives the gun a chance to fail: once per
  ``ko_check_period`` seconds it goes ``ko`` with probability
  ``ko_probability`` (``:14-20``; 600 s and 5 %).

Differences, on purpose:

* ``/nerf`` answers ``application/json`` ``{"status": "busy", "message":
  "shots:<N>", "x": .., "y": .., "shot": N}``.  The reference's simulator
  answers ``text/plain`` ``Nerf activated: ...`` (``:62-66``), which its own
  client cannot parse (``nerf_controller.py:17-21``), so every simulated fire
  there comes back as an error;
* state lives in an object guarded by a lock, not in a module global written
  from several threads (reference ``:11, 48, 72-74``), so several simulators
  can run in one process (tests, the bench);
* there is no thread that sleeps through the actuation (reference ``:69-74``).
  The gun records when the actuation ends and every look at its state first
  checks that time against the clock.  A ``/stop`` or a fault therefore ends
  an actuation for good -- nothing is left behind that could mark a later one
  idle -- and a reply that cannot be written (client gone) cannot leave the
  gun ``busy`` for ever;
* a non-integer ``shot`` is answered 400; the reference handler dies on the
  ``int()`` at ``:60`` and drops the connection with the gun stuck ``busy``;
* a ``POST`` whose ``Content-Length`` is missing reads as an empty form and
  one that is not a non-negative integer is answered 400; the reference
  handler dies on the ``int()`` at ``:38`` in both cases;
* a negative ``shot`` keeps the gun busy for no time at all; in the reference
  ``time.sleep`` of a negative number raises inside the actuation thread
  (``:73``) and the gun stays ``busy`` until ``/stop``;
* the clock and the random source can be injected; with an injected clock the
  gun's time only moves when the caller moves it, so its tests need no real
  waiting.  The port may be 0 (ephemeral).

Run as ``python -m nerf_amd.simulator [-p PORT]`` (reference ``:98-103``).
"""

import argparse
import json
import random
import threading
import time
from http.server import BaseHTTPRequestHandler, ThreadingHTTPServer
from urllib.parse import parse_qs, urlparse

IDLE, BUSY, KO = "idle", "busy", "ko"


class GunState:
    """The gun's three-word state machine; every transition holds ``_lock``.

    ``clock`` is any callable returning seconds; it times both the actuation
    and the fault roll.  ``busy`` ends by itself once the clock reaches the
    end of the actuation, which is checked whenever the state is looked at.
    """

    def __init__(self, seconds_per_shot=0.5, ko_probability=0.05, ko_check_period=600.0,
                 rng=None, clock=time.monotonic):
        self.seconds_per_shot = seconds_per_shot
        self.ko_probability = ko_probability
        self.ko_check_period = ko_check_period
        self._rng = rng if rng is not None else random.Random()
        self._clock = clock
        self._lock = threading.Lock()
        self._status = IDLE
        self._busy_until = 0.0  # meaningful while _status is BUSY
        self._last_ko_check = clock()

    def _current(self):
        """The state as of now; call with ``_lock`` held."""
        if self._status == BUSY and self._clock() >= self._busy_until:
            self._status = IDLE
        return self._status

    @property
    def status(self):
        with self._lock:
            return self._current()

    def busy_for(self):
        """Seconds of actuation left, 0 unless busy."""
        with self._lock:
            if self._current() != BUSY:
                return 0.0
            return self._busy_until - self._clock()

    def force(self, status, seconds=None):
        """Put the gun in a given state (tests and fault injection).

        A forced ``busy`` lasts ``seconds``; with ``seconds=None`` it lasts
        until ``/stop``, which is how the reference's gun ends up after a
        negative sleep or a dead handler (``simulator.py:60, 73``).
        ``seconds`` has no meaning for ``idle`` and ``ko``.
        """
        with self._lock:
            self._status = status
            self._busy_until = float("inf") if seconds is None else self._clock() + seconds

    def roll_ko(self):
        with self._lock:
            now = self._clock()
            if now - self._last_ko_check >= self.ko_check_period:
                if self._rng.random() < self.ko_probability:
                    self._status = KO
                self._last_ko_check = now

    def begin(self, shots):
        """Try to start an actuation of ``shots`` darts; the HTTP status that
        answers the request: 200 started, 429 busy, 503 ko."""
        with self._lock:
            state = self._current()
            if state == KO:
                return 503
            if state == BUSY:
                return 429
            self._status = BUSY
            self._busy_until = self._clock() + self.seconds_per_shot * shots
            return 200

    def stop(self):
        self.force(IDLE)


class _Handler(BaseHTTPRequestHandler):
    server_version = "NerfSim/1"

    def log_message(self, fmt, *args):  # quiet: one line per poll is noise
        pass

    @property
    def gun(self):
        return self.server.gun

    def do_GET(self):
        self.gun.roll_ko()
        target = urlparse(self.path)
        if target.path == "/nerf":
            self._nerf(parse_qs(target.query))
        elif target.path == "/stop":
            self.gun.stop()
            self._reply(200, "text/plain", b"Nerf stopped")
        elif target.path == "/status":
            self._reply(200, "application/json", json.dumps({"status": self.gun.status}).encode())
        else:
            self.send_error(404, "Not Found")

    def do_POST(self):
        self.gun.roll_ko()
        try:
            length = int(self.headers.get("Content-Length") or 0)
            if length < 0:
                raise ValueError(length)
        except ValueError:
            # the body's extent is unknown, so the connection cannot be reused
            self.close_connection = True
            self.send_error(400, "Content-Length must be a non-negative integer")
            return
        form = parse_qs(self.rfile.read(length).decode("utf-8", "replace"))
        if self.path == "/nerf":
            self._nerf(form)
        else:
            self.send_error(404, "Not Found")

    def _nerf(self, fields):
        x = fields.get("x", ["0"])[0]
        y = fields.get("y", ["0"])[0]
        try:
            shot = int(fields.get("shot", ["1"])[0])
        except ValueError:
            self.send_error(400, "shot must be an integer")
            return
        code = self.gun.begin(shot)
        if code == 503:
            self.send_error(503, "Service Unavailable")
            return
        if code == 429:
            self.send_error(429, "Too Many Requests")
            return
        body = {"status": BUSY, "message": f"shots:{shot}", "x": x, "y": y, "shot": shot}
        self._reply(200, "application/json", json.dumps(body).encode())

    def _reply(self, code, content_type, payload):
        self.send_response(code)
        self.send_header("Content-Type", content_type)
        self.send_header("Content-Length", str(len(payload)))
        self.end_headers()
        self.wfile.write(payload)


class GunSimulator:
    """One simulated gun behind an HTTP server on ``host:port``.

    Usable as a context manager; ``url`` is what to hand to
    :class:`~nerf_amd.nerf_controller.NerfController`.
    """

    def __init__(self, port=5555, host="127.0.0.1", **gun_options):
        self.gun = GunState(**gun_options)
        self._httpd = ThreadingHTTPServer((host, port), _Handler)
        self._httpd.daemon_threads = True
        self._httpd.gun = self.gun
        self._thread = None

    @property
    def port(self):
        return self._httpd.server_address[1]

    @property
    def url(self):
        """Base URL of the bound address; loopback when bound to all addresses."""
        host = self._httpd.server_address[0]
        if host in ("", "0.0.0.0"):
            host = "127.0.0.1"
        return f"http://{host}:{self.port}"

    def start(self):
        # short poll: close() returns within this interval
        self._thread = threading.Thread(target=self._httpd.serve_forever, args=(0.02,), daemon=True)
        self._thread.start()
        return self

    def serve_forever(self):
        self._httpd.serve_forever()

    def close(self):
        self.gun.stop()
        if self._thread is not None:
            self._httpd.shutdown()
            self._thread.join()
            self._thread = None
        self._httpd.server_close()

    def __enter__(self):
        return self.start()

    def __exit__(self, *exc):
        self.close()


def main(argv=None):
    parser = argparse.ArgumentParser(description="Nerf gun server simulator")
    parser.add_argument("-p", "--port", type=int, default=5555,
                        help="port to listen on (default: 5555)")
    parser.add_argument("--host", default="", help="address to bind (default: all)")
    parser.add_argument("--seconds-per-shot", type=float, default=0.5)
    args = parser.parse_args(argv)
    sim = GunSimulator(port=args.port, host=args.host, seconds_per_shot=args.seconds_per_shot)
    print(f"Nerf simulator listening on port {sim.port}", flush=True)
    try:
        sim.serve_forever()
    except KeyboardInterrupt:
        pass
    finally:
        sim.close()


if __name__ == "__main__":
    main()
