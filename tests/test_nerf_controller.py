"""NerfController against the reference's own known answers, plus the edges of
reference nerf_controller.py:19-21 and :56-67 that its tests leave open."""

import json
import os
from unittest.mock import Mock, patch

import pytest

from nerf_amd.nerf_controller import NerfController, _shots_from_message

HERE = os.path.dirname(__file__)
with open(os.path.join(HERE, "golden", "nerf_controller_kat.json")) as fh:
    KAT = json.load(fh)
CASES = {c["name"]: c for c in KAT["cases"]}

GET = "nerf_amd.nerf_controller.requests.get"
WAIT = "nerf_amd.nerf_controller.NerfController.wait_until_idle"
STATUS = "nerf_amd.nerf_controller.NerfController.get_status"


@pytest.fixture
def controller():
    return NerfController(KAT["server_url"])


class _FakeTime:
    """Stands in for the ``time`` module inside the client: ``sleep`` moves
    ``monotonic`` forward at once, so polling tests take no real time.  The
    wall clock steps back an hour at every reading: a wait timed on it
    (reference nerf_controller.py:53-54) would never end."""

    def __init__(self):
        self.now = 5000.0
        self.wall = 1.7e9
        self.slept = []

    def monotonic(self):
        return self.now

    def time(self):
        self.wall -= 3600.0
        return self.wall

    def sleep(self, seconds):
        self.slept.append(seconds)
        self.now += seconds


@pytest.fixture
def fake_time():
    fake = _FakeTime()
    with patch("nerf_amd.nerf_controller.time", fake):
        yield fake


def _response(spec):
    r = Mock()
    r.raise_for_status.return_value = None
    if "json" in spec:
        r.json.return_value = spec["json"]
    if "text" in spec:
        r.text = spec["text"]
    return r


def _as_returned(value):
    """JSON has no tuples; the client returns (ok, info) pairs."""
    return tuple(value) if isinstance(value, list) else value


@pytest.mark.parametrize(
    "name", ["fire_success", "fire_error", "stop_success", "get_status_success", "get_status_error"]
)
def test_transport_known_answers(name, controller):
    case = CASES[name]
    expect = case["expect"]
    with patch(GET) as get, patch(WAIT) as wait:
        if "raises" in case["transport"]:
            get.side_effect = Exception(case["transport"]["raises"])
        else:
            get.return_value = _response(case["transport"])
        if "wait_until_idle_returns" in case:
            wait.return_value = tuple(case["wait_until_idle_returns"])
        result = getattr(controller, case["call"]["method"])(**case["call"]["kwargs"])

        assert result == _as_returned(expect["result"])
        if "get_called_once_with" in expect:
            spec = expect["get_called_once_with"]
            get.assert_called_once_with(*spec["args"], **spec["kwargs"])
        if "wait_called_once_with" in expect:
            spec = expect["wait_called_once_with"]
            wait.assert_called_once_with(*spec["args"], **spec["kwargs"])


def test_wait_until_idle_known_answer_success(controller, fake_time):
    case = CASES["wait_until_idle_success"]
    with patch(STATUS) as get_status:
        get_status.side_effect = [dict(s) for s in case["get_status_sequence"]]
        ok, status = controller.wait_until_idle(**case["call"]["kwargs"])
        assert ok is case["expect"]["ok"]
        assert status["status"] == case["expect"]["final_status"]
        assert get_status.call_count == case["expect"]["get_status_calls"]
    assert fake_time.slept == [0.1, 0.1]  # one check_interval after each busy poll, reference :68


def test_wait_until_idle_known_answer_timeout(controller, fake_time):
    case = CASES["wait_until_idle_timeout"]
    with patch(STATUS) as get_status:
        get_status.return_value = dict(case["get_status_always"])
        ok, status = controller.wait_until_idle(**case["call"]["kwargs"])
        assert ok is case["expect"]["ok"]
        assert status["status"] == case["expect"]["final_status"]
        assert get_status.call_count > case["expect"]["get_status_calls_more_than"]
        # polls at 0, 0.1, ... while less than 0.5 s have passed (reference :54)
        assert get_status.call_count in (5, 6)
    assert sum(fake_time.slept) >= 0.5


def test_wait_until_idle_known_answers_in_real_time(controller):
    # the same two cases on the real clock, as the reference's tests run them
    with patch(STATUS) as get_status:
        get_status.side_effect = [dict(s) for s in CASES["wait_until_idle_success"]["get_status_sequence"]]
        ok, status = controller.wait_until_idle(timeout=5, check_interval=0.01)
        assert ok is True and status["status"] == "idle" and get_status.call_count == 3
    with patch(STATUS) as get_status:
        get_status.return_value = {"status": "busy"}
        ok, status = controller.wait_until_idle(timeout=0.05, check_interval=0.01)
        assert ok is False and status["status"] == "busy" and get_status.call_count > 1


# ---- edges the reference's tests do not pin; expectations follow the cited lines ----


def test_trailing_slash_is_dropped():
    # reference nerf_controller.py:9
    assert NerfController("http://h:1/").server_url == "http://h:1"
    assert NerfController("http://h:1///").server_url == "http://h:1"


@pytest.mark.parametrize(
    "message, shots",
    [
        ("shots:3", 3),
        ("fired shots:12", 12),
        ("shots: 4 ", 4),  # int() tolerates surrounding blanks
        ("shots:2 shots:9", 2),  # text between the first two tags
        ("no tag here", -1),
        ("", -1),
    ],
)
def test_shots_from_message(message, shots):
    # reference nerf_controller.py:19-21
    assert _shots_from_message(message) == shots


def test_shots_tag_followed_by_text_is_an_error(controller):
    # int("3 darts") raises inside fire's try block -> error tuple (reference :21, :27-29)
    with patch(GET) as get:
        get.return_value = _response({"json": {"message": "shots:3 darts"}})
        ok, info = controller.fire(1, 2, 3)
    assert ok is False
    assert info["status"] == "error" and info["shots"] == 0


def test_fire_reply_without_message_means_zero_shots(controller):
    # reference :19 -- default "shots:0"
    with patch(GET) as get, patch(WAIT) as wait:
        get.return_value = _response({"json": {}})
        wait.return_value = (True, {"status": "idle", "shots": 0})
        controller.fire(1, 2, 3)
        wait.assert_called_once_with(shots=0)


def test_fire_without_wait_returns_the_reply(controller):
    # reference :23-26
    with patch(GET) as get, patch(STATUS) as get_status:
        get.return_value = _response({"json": {"message": "shots:0", "extra": 1}})
        assert controller.fire(90, 45, 0, False) == (True, {"message": "shots:0", "extra": 1})
        get_status.assert_not_called()


def test_fire_http_error_status_is_an_error(controller):
    # raise_for_status inside the try block (reference :16)
    with patch(GET) as get:
        r = _response({"json": {}})
        r.raise_for_status.side_effect = Exception("429 Client Error: Too Many Requests")
        get.return_value = r
        assert controller.fire() == (
            False,
            {"status": "error", "message": "429 Client Error: Too Many Requests", "shots": 0},
        )


def test_fire_defaults(controller):
    # reference :11 -- x=0, y=0, shot=1, wait=True
    with patch(GET) as get, patch(WAIT) as wait:
        get.return_value = _response({"json": {"message": "shots:1"}})
        wait.return_value = (True, {"status": "idle", "shots": 1})
        controller.fire()
        get.assert_called_once_with(
            "http://test-server.com/nerf", params={"x": 0, "y": 0, "shot": 1}
        )


def test_stop_error_is_text(controller):
    # reference :37-38
    with patch(GET) as get:
        get.side_effect = Exception("boom")
        assert controller.stop() == "Error: boom"


@pytest.mark.parametrize(
    "status, shots_arg, ok, shots_out",
    [
        ({"status": "idle"}, 4, True, 4),  # reference :56-59
        ({"status": "idle", "shots": 2}, 4, True, 2),  # server's count wins
        ({"status": "ko"}, 4, False, 0),  # reference :60-63
        ({"status": "ko", "shots": 1}, 4, False, 1),
        ({"status": "error", "message": "x"}, 4, False, 0),  # reference :64-67
    ],
)
def test_wait_until_idle_fills_in_shots(controller, fake_time, status, shots_arg, ok, shots_out):
    with patch(STATUS) as get_status:
        get_status.return_value = dict(status)
        got_ok, got = controller.wait_until_idle(timeout=1, check_interval=0.01, shots=shots_arg)
    assert got_ok is ok
    assert got["shots"] == shots_out
    assert got["status"] == status["status"]


def test_wait_until_idle_timeout_keeps_last_status_untouched(controller, fake_time):
    # reference :69 -- the busy status is returned as it came, no "shots" added
    with patch(STATUS) as get_status:
        get_status.return_value = {"status": "busy"}
        ok, status = controller.wait_until_idle(timeout=0.05, check_interval=0.01, shots=7)
    assert ok is False and status == {"status": "busy"}


def test_wait_until_idle_expired_before_first_poll(controller):
    # the reference raises here (status unbound at :69); this client reports an error
    with patch(STATUS) as get_status:
        ok, status = controller.wait_until_idle(timeout=0)
        get_status.assert_not_called()
    assert ok is False and status["status"] == "error" and status["shots"] == 0


def test_no_method_raises_on_unreachable_server():
    c = NerfController("http://127.0.0.1:9")  # discard port: connection refused
    ok, info = c.fire(1, 2, 3)
    assert ok is False and info["status"] == "error" and info["shots"] == 0
    assert c.get_status()["status"] == "error"
    assert c.stop().startswith("Error: ")


def test_request_timeout_is_passed_only_when_set():
    with patch(GET) as get:
        get.return_value = _response({"json": {"status": "idle"}, "text": "Nerf stopped"})
        c = NerfController("http://h:1", request_timeout=2.5)
        c.get_status()
        get.assert_called_with("http://h:1/status", timeout=2.5)
        c.stop()
        get.assert_called_with("http://h:1/stop", timeout=2.5)
        c.fire(1, 2, 3, wait=False)
        get.assert_called_with("http://h:1/nerf", params={"x": 1, "y": 2, "shot": 3}, timeout=2.5)
        NerfController("http://h:1").get_status()
        get.assert_called_with("http://h:1/status")  # the shape the reference's tests pin


def test_request_timeout_bounds_a_server_that_never_answers():
    import socket

    with socket.socket() as listener:  # accepts the connection, never replies
        listener.bind(("127.0.0.1", 0))
        listener.listen(1)
        c = NerfController(f"http://127.0.0.1:{listener.getsockname()[1]}", request_timeout=0.2)
        status = c.get_status()
    assert status["status"] == "error" and "timed out" in status["message"].lower()


def test_wait_until_idle_survives_malformed_status_replies(controller, fake_time):
    # the reference raises KeyError / TypeError at nerf_controller.py:56
    with patch(STATUS) as get_status:
        get_status.side_effect = [{}, {"state": "idle"}, {"status": "idle"}]
        assert controller.wait_until_idle(timeout=5, shots=2) == (True, {"status": "idle", "shots": 2})
        assert get_status.call_count == 3
    with patch(STATUS) as get_status:
        get_status.return_value = ["idle"]
        ok, status = controller.wait_until_idle(timeout=5)
        assert ok is False and status["status"] == "error" and status["shots"] == 0
        assert get_status.call_count == 1
    with patch(STATUS) as get_status:
        get_status.return_value = {"no": "status"}
        ok, status = controller.wait_until_idle(timeout=0.3, check_interval=0.1)
        assert ok is False and status == {"no": "status"}
