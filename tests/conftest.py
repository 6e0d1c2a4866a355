import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# Under `pytest -x` one peripheral failure must not leave the hot path's parity unreached: the files that carry
# SURVEY section 8(a)'s rows (decoder.ml:142-149, 213-224, 347-360; dct.ml:11-107; encoder.ml:81-108) run first, in this
# order; everything else keeps pytest's own (alphabetical) order behind them.
HOT_PATH_FIRST = ("test_gpu_decode", "test_gpu_encode_upsample", "test_gpu_jpeg_api", "test_cpp_model", "test_gpu_config1_128", "test_gpu_full_configs",
                  "test_gpu_yuv444", "test_gpu_huffman", "test_gpu_hdec")


def pytest_collection_modifyitems(config, items):
    rank = {name: i for i, name in enumerate(HOT_PATH_FIRST)}

    def key(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return rank.get(mod, len(rank))
    items.sort(key=key)   # stable: order inside a file, and among the rest, is unchanged


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def golden_json(name):
    import json
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def golden_bytes(name):
    with open(os.path.join(GOLDEN, name), "rb") as f:
        return f.read()
