"""The simulator over real loopback HTTP, driven through NerfController.

The four end-to-end behaviours are those of the reference's live-server tests
(nerf-gun-control/test/test_nerf_controller_integration.py:21-83): fire then
idle, stop then idle, three fires in a row, rejection while busy.  There they
need a server on localhost:5555 and are skipped by default; here the server is
in-process on an ephemeral port with a short actuation time.
"""

import os
import random
import re
import subprocess
import sys
import threading

import requests

from nerf_amd import GunSimulator, NerfController
from nerf_amd.simulator import BUSY, IDLE, KO, GunState


def test_fire_and_wait(simulator):
    c = NerfController(simulator.url)
    ok, status = c.fire(x=10, y=20, shot=3)
    assert ok is True
    assert status == {"status": "idle", "shots": 3}
    ok, _ = c.wait_until_idle(timeout=5)
    assert ok
    assert c.get_status() == {"status": "idle"}


def test_stop(simulator):
    c = NerfController(simulator.url)
    assert "Nerf stopped" in c.stop()
    assert c.get_status()["status"] == "idle"


def test_multiple_fires(simulator):
    c = NerfController(simulator.url)
    for i in range(3):
        ok, status = c.fire(x=i * 10, y=i * 5, shot=i + 1)
        assert ok is True and status["shots"] == i + 1
        assert c.get_status()["status"] == "idle"


def test_busy_state(frozen_simulator, clock):
    c = NerfController(frozen_simulator.url)
    ok, reply = c.fire(x=0, y=0, shot=10, wait=False)
    assert ok is True
    assert reply["message"] == "shots:10" and reply["status"] == "busy"
    assert c.get_status()["status"] == "busy"
    busy_ok, busy_info = c.fire(x=10, y=10, shot=1)
    assert not busy_ok
    assert "429" in busy_info["message"] and busy_info["shots"] == 0
    clock.advance(4.99)  # 10 shots at 0.5 s: reference simulator.py:73
    assert c.get_status()["status"] == "busy"
    clock.advance(0.01)
    ok, status = c.wait_until_idle(timeout=10, shots=10)
    assert ok and status == {"status": "idle", "shots": 10}


def test_nerf_reply_is_json_the_client_can_parse(simulator):
    r = requests.get(simulator.url + "/nerf", params={"x": -5, "y": 7, "shot": 2})
    assert r.status_code == 200
    assert r.headers["Content-Type"] == "application/json"
    assert r.json() == {"status": "busy", "message": "shots:2", "x": "-5", "y": "7", "shot": 2}


def test_nerf_defaults(simulator):
    # reference simulator.py:58-60 -- x=0, y=0, shot=1 when absent
    body = requests.get(simulator.url + "/nerf").json()
    assert (body["x"], body["y"], body["shot"]) == ("0", "0", 1)


def test_post_nerf_form(simulator):
    # reference simulator.py:36-45
    r = requests.post(simulator.url + "/nerf", data={"x": 1, "y": 2, "shot": 1})
    assert r.status_code == 200 and r.json()["message"] == "shots:1"
    assert requests.post(simulator.url + "/status", data={}).status_code == 404


def test_unknown_path_is_404(simulator):
    assert requests.get(simulator.url + "/reload").status_code == 404


def test_bad_shot_is_400_and_gun_stays_idle(simulator):
    assert requests.get(simulator.url + "/nerf", params={"shot": "two"}).status_code == 400
    assert simulator.gun.status == IDLE


def test_zero_shots_moves_and_goes_idle_at_once(simulator):
    # return-to-home is fire(home_x, home_y, 0, wait=False): reference main.py:784-786
    c = NerfController(simulator.url)
    ok, reply = c.fire(90, 45, 0, False)
    assert ok and reply["message"] == "shots:0"
    ok, status = c.wait_until_idle(timeout=2)
    assert ok and status["shots"] == 0


def test_ko_gun_answers_503_and_client_reports_ko(simulator):
    c = NerfController(simulator.url)
    simulator.gun.force(KO)
    assert requests.get(simulator.url + "/nerf").status_code == 503
    ok, info = c.fire(1, 1, 1)
    assert ok is False and "503" in info["message"]
    assert c.wait_until_idle(timeout=1) == (False, {"status": "ko", "shots": 0})
    # /stop revives it, as in the reference (simulator.py:76-78)
    c.stop()
    assert c.get_status() == {"status": "idle"}


def test_gun_going_ko_mid_actuation_fails_the_fire(frozen_simulator, clock):
    c = NerfController(frozen_simulator.url)
    polls = []

    def break_on_third_poll(original=c.get_status):
        polls.append(1)
        if len(polls) == 3:
            frozen_simulator.gun.force(KO)
        return original()

    c.get_status = break_on_third_poll
    ok, status = c.fire(0, 0, 5)
    assert ok is False and status == {"status": "ko", "shots": 0}
    assert len(polls) == 3
    clock.advance(60)  # long past the actuation's natural end: still ko
    assert frozen_simulator.gun.status == KO


def test_stop_cuts_an_actuation_short_and_it_cannot_idle_a_later_one(frozen_simulator, clock):
    c = NerfController(frozen_simulator.url)
    gun = frozen_simulator.gun
    c.fire(0, 0, 2, wait=False)  # would end at t+1.0
    clock.advance(0.25)
    assert c.stop() == "Nerf stopped"
    assert gun.status == IDLE
    c.fire(0, 0, 5, wait=False)  # ends at t+0.25+2.5
    clock.advance(1.0)  # past the first actuation's natural end
    assert gun.status == BUSY
    clock.advance(1.49)
    assert gun.status == BUSY
    clock.advance(0.01)
    assert gun.status == IDLE
    assert c.wait_until_idle(timeout=2)[0]


def test_actuation_lasts_seconds_per_shot_times_shots(clock):
    # reference simulator.py:73 -- sleep(0.5 * shots)
    gun = GunState(ko_probability=0.0, clock=clock)
    assert gun.seconds_per_shot == 0.5
    assert gun.begin(3) == 200
    assert gun.status == BUSY and gun.busy_for() == 1.5
    assert gun.begin(1) == 429  # reference :52-54
    clock.advance(1.499)
    assert gun.status == BUSY
    clock.advance(0.001)
    assert gun.status == IDLE and gun.busy_for() == 0.0
    assert gun.begin(1) == 200


def test_ko_gun_refuses_until_stopped(clock):
    # reference simulator.py:49-51 and :76-78
    gun = GunState(ko_probability=0.0, clock=clock)
    gun.force(KO)
    assert gun.begin(1) == 503
    clock.advance(3600)
    assert gun.status == KO
    gun.stop()
    assert gun.status == IDLE and gun.begin(1) == 200


def test_zero_and_negative_shots_hold_the_gun_for_no_time(clock):
    gun = GunState(ko_probability=0.0, clock=clock)
    assert gun.begin(0) == 200 and gun.status == IDLE
    assert gun.begin(-2) == 200 and gun.status == IDLE


def test_forced_busy_lasts_as_long_as_asked(clock):
    gun = GunState(ko_probability=0.0, clock=clock)
    gun.force(BUSY, seconds=2.0)
    assert gun.status == BUSY
    clock.advance(2.0)
    assert gun.status == IDLE


def test_forced_busy_without_a_duration_lasts_until_stop(clock):
    # the state the reference's gun is left in by a dead handler or a negative
    # sleep (simulator.py:60, 73): busy until /stop
    gun = GunState(ko_probability=0.0, clock=clock)
    gun.force(BUSY)
    clock.advance(86400)
    assert gun.status == BUSY and gun.begin(1) == 429
    gun.stop()
    assert gun.status == IDLE and gun.begin(1) == 200


def test_reply_that_cannot_be_written_does_not_leave_the_gun_busy_for_ever(frozen_simulator, clock):
    # the /nerf reply fails to go out as it does when the client has gone
    # away; the handler dies after the actuation began, and the actuation
    # still ends by the clock
    from unittest.mock import patch

    from nerf_amd.simulator import _Handler

    attempts = []

    def client_gone(self, *args):
        attempts.append(args[0])
        raise BrokenPipeError("client went away")

    # socketserver prints the traceback of a handler that dies; keep it quiet
    with patch.object(_Handler, "_reply", client_gone), \
            patch.object(frozen_simulator._httpd, "handle_error", lambda *a: None):
        try:
            requests.get(frozen_simulator.url + "/nerf", params={"shot": 2}, timeout=5)
        except requests.RequestException:
            pass  # the connection was dropped without a reply
    assert attempts == [200]
    assert frozen_simulator.gun.status == BUSY
    clock.advance(0.99)
    assert frozen_simulator.gun.status == BUSY
    clock.advance(0.01)
    assert frozen_simulator.gun.status == IDLE
    assert NerfController(frozen_simulator.url).fire(0, 0, 1, wait=False)[0] is True


def _raw_request(port, payload):
    import socket

    with socket.create_connection(("127.0.0.1", port), timeout=5) as sock:
        sock.sendall(payload)
        chunks = []
        while True:
            chunk = sock.recv(4096)
            if not chunk:
                break
            chunks.append(chunk)
    return b"".join(chunks).decode("latin-1")


def test_post_with_unreadable_content_length_is_400_and_gun_stays_idle(simulator):
    # the reference handler dies on int() of the header: simulator.py:38
    for value in (b"abc", b"-1", b"1e3"):
        reply = _raw_request(
            simulator.port,
            b"POST /nerf HTTP/1.1\r\nHost: x\r\nContent-Length: " + value + b"\r\n\r\nshot=3")
        assert reply.startswith("HTTP/1.0 400 ") or reply.startswith("HTTP/1.1 400 "), reply[:80]
        assert simulator.gun.status == IDLE


def test_post_without_content_length_reads_as_an_empty_form(simulator):
    reply = _raw_request(simulator.port, b"POST /nerf HTTP/1.0\r\nHost: x\r\n\r\n")
    assert " 200 " in reply.splitlines()[0]
    assert '"message": "shots:1"' in reply  # the defaults of reference simulator.py:58-60


def test_url_names_the_bound_address():
    with GunSimulator(port=0, host="127.0.0.2", ko_probability=0.0) as sim:
        assert sim.url == f"http://127.0.0.2:{sim.port}"
        assert NerfController(sim.url).get_status() == {"status": "idle"}
    for everywhere in ("", "0.0.0.0"):
        with GunSimulator(port=0, host=everywhere, ko_probability=0.0) as sim:
            assert sim.url == f"http://127.0.0.1:{sim.port}"
            assert NerfController(sim.url).get_status() == {"status": "idle"}


def test_ko_roll_period_and_probability():
    # reference simulator.py:14-20: at most one roll per period, ko when roll < p
    now = [0.0]
    rolls = iter([0.5, 0.01])

    class Rng:
        def random(self):
            return next(rolls)

    gun = GunState(ko_probability=0.05, ko_check_period=600, rng=Rng(), clock=lambda: now[0])
    gun.roll_ko()  # period not over: no roll consumed
    assert gun.status == IDLE
    now[0] = 600.0
    gun.roll_ko()  # roll 0.5 >= 0.05
    assert gun.status == IDLE
    now[0] = 900.0
    gun.roll_ko()  # only 300 s since the last roll
    assert gun.status == IDLE
    now[0] = 1200.0
    gun.roll_ko()  # roll 0.01 < 0.05
    assert gun.status == KO


def test_ko_never_when_probability_zero():
    now = [0.0]
    gun = GunState(ko_probability=0.0, ko_check_period=1, rng=random.Random(0), clock=lambda: now[0])
    for t in range(1, 200):
        now[0] = float(t)
        gun.roll_ko()
    assert gun.status == IDLE


def test_two_simulators_are_independent(clock):
    with GunSimulator(port=0, ko_probability=0.0, clock=clock) as a, \
            GunSimulator(port=0, ko_probability=0.0, clock=clock) as b:
        NerfController(a.url).fire(0, 0, 3, wait=False)
        assert a.gun.status == BUSY and b.gun.status == IDLE
        assert NerfController(b.url).fire(0, 0, 1, wait=False)[0] is True


def test_concurrent_fires_admit_exactly_one(frozen_simulator):
    codes = []

    def shoot():
        codes.append(requests.get(frozen_simulator.url + "/nerf", params={"shot": 1}).status_code)

    threads = [threading.Thread(target=shoot) for _ in range(8)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert sorted(codes) == [200] + [429] * 7


def test_command_line_starts_a_server_on_the_given_port():
    # reference simulator.py:98-103: -p/--port, default 5555; 0 asks for any free port
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    proc = subprocess.Popen(
        [sys.executable, "-m", "nerf_amd.simulator", "-p", "0", "--host", "127.0.0.1",
         "--seconds-per-shot", "0.01"],
        cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        banner = proc.stdout.readline()
        port = int(re.fullmatch(r"Nerf simulator listening on port (\d+)\n", banner).group(1))
        assert port > 0
        c = NerfController(f"http://127.0.0.1:{port}")
        assert c.get_status() == {"status": "idle"}
        assert c.fire(1, 2, 2) == (True, {"status": "idle", "shots": 2})
    finally:
        proc.terminate()
        proc.wait(timeout=10)


def test_command_line_arguments():
    from unittest.mock import patch

    from nerf_amd import simulator as module

    with patch.object(module, "GunSimulator") as cls:
        cls.return_value.port = 5555
        module.main([])
        cls.assert_called_once_with(port=5555, host="", seconds_per_shot=0.5)
        cls.return_value.serve_forever.assert_called_once_with()
        cls.return_value.close.assert_called_once_with()
    with patch.object(module, "GunSimulator") as cls:
        cls.return_value.port = 6001
        cls.return_value.serve_forever.side_effect = KeyboardInterrupt
        module.main(["--port", "6001"])  # Ctrl-C ends it quietly and closes the socket
        assert cls.call_args.kwargs["port"] == 6001
        cls.return_value.close.assert_called_once_with()
