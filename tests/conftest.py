import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Build what can be built where we are (hipcc / gcc exist here and on the GPU box; the compiled reference
    # under oracle/_ref and the drop-in under integration/_build need /root/reference, i.e. this container --
    # on the GPU box the prebuilt files that travelled with the snapshot are used).
    import subprocess

    for d in ("minizip-ng_amd/csrc", "oracle", "integration"):
        subprocess.run(["make", "-s", "-C", os.path.join(ROOT, d)], capture_output=True)


@pytest.fixture(scope="session")
def fixtures():
    import base64
    import json

    with open(os.path.join(ROOT, "tests", "golden", "fixtures.json")) as f:
        ents = json.load(f)["entries"]
    for e in ents:
        e["payload"] = base64.b64decode(e["payload"])
    return ents
