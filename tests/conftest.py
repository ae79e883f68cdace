import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def cfg():
    from radzero_amd.config import RadZeroConfig
    return RadZeroConfig()


@pytest.fixture(scope="session")
def state_dict(cfg):
    """The synthetic checkpoint the goldens were generated from (regenerated, never stored)."""
    from radzero_amd.weights import make_state_dict
    return make_state_dict(cfg, 20260103)


@pytest.fixture(scope="session")
def oracle(cfg, state_dict):
    from oracle.radzero_oracle import OracleModel
    return OracleModel(state_dict, cfg, attn_impl="eager")


def load_golden(name):
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False))


GOLDEN_CASES = ["g1_s224_b1_t1", "g2_s224_b2_t3", "g2_s224_b1_t14", "g2_s224_b3_t1",
                "g3_s266_b2_t3", "g3_s518_b1_t14", "g5_s224_b1_t64_l32", "g7_s1024_b1_t14"]


def post_map_cases():
    """tests/golden/post_maps.npz (tools/make_goldens_post.py: the reference's own interpolate_similarity_scores /
    get_grounding_point): yields (golden_name, (H, W), keep_aspect_ratio, moments (T,3) f64, samples (T,n) f32, points (T,2))."""
    import numpy as np
    z = np.load(os.path.join(GOLDEN_DIR, "post_maps.npz"), allow_pickle=False)
    stride = int(z["sample_stride"])
    for key in [str(k) for k in z["cases"]]:
        gname, size, proc = key.split("|")
        h, w = (int(v) for v in size.split("x"))
        yield gname, (h, w), proc == "aspect", z[key + "|moments"], z[key + "|samples"], z[key + "|points"], stride
