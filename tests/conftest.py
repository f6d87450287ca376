import os
import sys

import pytest

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))


def pytest_configure(config):
    config.addinivalue_line(
        "markers",
        "gpu: needs an MI355X. No test carries this marker: the package has no "
        "device code because the reference has no data-parallel path.",
    )


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if (config.getoption("-m") or "").strip() == "gpu":
        terminalreporter.write_line(
            "no GPU tests exist by design: Vadiml1024/nerf is a chat bot driving a "
            "dart gun over HTTP; nerf_amd ships no HIP code, so '-m gpu' selects nothing."
        )


@pytest.fixture
def simulator():
    """A gun simulator on an ephemeral loopback port, 20 ms per shot, no random faults."""
    from nerf_amd.simulator import GunSimulator

    with GunSimulator(port=0, seconds_per_shot=0.02, ko_probability=0.0) as sim:
        yield sim
