"""The outlier-channel checkpoint (weights.add_outlier_channels: residual |max| ~ 480, every LayerNorm dominated by four channels — what
trained DINOv2 weights look like) at the shapes bench.py TIMES: G14 = one 1024^2 image x 14 prompts (N = 5330, the
headline) and G15 = two 518^2 images x 14 prompts (N = 1370, the released resolution), outputs of the reference itself
(tools/make_goldens_post.py --outlier-timed-shapes; exp/cxr_pt/model/modeling.py:96-123, :278-328, losses.py:187-240).

The fp32 mode must hold north_star's 1e-3 with DEFAULT options (the MX form: e4m3 correction planes, the arithmetic `value_1e3_mode` measures; round 6:
taken wherever a launch's rows are a multiple of 256, i.e. alone AND embedded in the batch the bench times — B = 32 at 1024^2, B = 64 at 518^2 — with the
SAME bits, 20 x more keys and rows feeding the planes than G8's N = 257) and on its other arithmetic, the three-plane f16 form (gemm_f32_mx = 0),
with exact class argmax, exact patch argmax wherever the reference's top-2 margin exceeds twice the measured error, and no guard re-run.
bf16 / fp16 stay finite and inside their stated gates (1.5 x the error measured at these shapes; max is heavy-tailed on this checkpoint,
the rms is the stable figure)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from radzero_amd.synthetic import synthetic_pixels

pytestmark = pytest.mark.gpu

FP32_TOL = 1e-3
CASES = {"g14_outlier_s1024_b1_t14": 32, "g15_outlier_s518_b2_t14": 64}       # golden -> the batch size bench.py times at that resolution


@pytest.fixture(scope="module")
def outlier_sd(cfg, state_dict):
    from radzero_amd.weights import add_outlier_channels
    return add_outlier_channels(state_dict, cfg)


def _golden_inputs(g):
    px = torch.from_numpy(synthetic_pixels(int(g["batch"]), int(g["side"]), int(g["px_seed"]))).cuda()
    enc = {"input_ids": torch.from_numpy(g["input_ids"]).cuda(), "attention_mask": torch.from_numpy(g["attention_mask"]).cuda()}
    return px, enc


def _errors(out, rows, g):
    sim = out["similarity_scores"][rows].cpu().numpy()
    lg = np.atleast_2d(out["logits"].cpu().numpy())[rows] if out["logits"].ndim == 2 else np.atleast_2d(out["logits"].cpu().numpy())
    ref_l = np.atleast_2d(g["logits"])
    d = sim - g["similarity_scores"]
    return float(np.abs(d).max()), float(np.sqrt((d * d).mean())), float(np.abs(lg - ref_l).max()), sim, lg


def _check_argmax(sim, lg, g, e_s):
    ref_l = np.atleast_2d(g["logits"])
    assert np.array_equal(lg.argmax(1), ref_l.argmax(1))                       # class index per image: reference margins 0.07 / 0.20
    ref = g["similarity_scores"]
    srt = np.sort(ref, -1)
    decided = (srt[..., -1] - srt[..., -2]) > 2.0 * e_s                        # a swap needs both errors to add up to the margin
    same = sim.argmax(-1) == ref.argmax(-1)
    assert same[decided].all(), (int((~same & decided).sum()), int(decided.sum()))
    return int(same.sum()), int(decided.sum()), same.size


@pytest.mark.parametrize("name", list(CASES))
def test_fp32_default_options_on_the_outlier_checkpoint_at_timed_shapes(name, cfg, outlier_sd):
    from radzero_amd.modeling import RadZeroModel
    g = load_golden(name)
    nb, side, big = int(g["batch"]), int(g["side"]), CASES[name]
    m = RadZeroModel.from_state_dict(outlier_sd, cfg, torch_dtype=torch.float32, device="cuda:0").eval()
    try:
        assert m.get_model_option("gemm_f32_mx") == 1 and m.get_model_option("attn_f32_mx") == 1 and m.get_model_option("attn_f32_pv") == 0
        assert m.get_model_option("f32_split_guard") == 1
        px, enc = _golden_inputs(g)
        rows = list(range(nb))
        # (a) the three-plane form (option gemm_f32_mx = 0), alone
        m.set_model_option("gemm_f32_mx", 0)
        out3 = m.compute_logits(px, [enc])
        assert m.get_model_option("last_f32_form") == 1
        e_s, r_s, e_l, sim, lg = _errors(out3, rows, g)
        assert np.isfinite(sim).all() and np.isfinite(lg).all()
        hits = _check_argmax(sim, lg, g, e_s)
        print(f"\n[{name} fp32, three-plane form, alone] max|dscores|={e_s:.2e} rms {r_s:.2e} max|dlogits|={e_l:.2e} patch argmax {hits}")
        assert e_s <= FP32_TOL and e_l <= FP32_TOL
        three = out3["similarity_scores"].clone()
        m.set_model_option("gemm_f32_mx", None)
        # (b) default options, alone: the MX form (round 6: wherever the rows are a multiple of 256)
        out = m.compute_logits(px, [enc])
        assert m.get_model_option("last_f32_form") == 2
        a_s, ar_s, a_l, sima, lga = _errors(out, rows, g)
        hitsa = _check_argmax(sima, lga, g, a_s)
        print(f"[{name} fp32 default (MX form), alone] max|dscores|={a_s:.2e} rms {ar_s:.2e} max|dlogits|={a_l:.2e} patch argmax {hitsa}")
        assert a_s <= FP32_TOL and a_l <= FP32_TOL
        alone = out["similarity_scores"].clone()
        # (c) inside the batch the bench times: the same form, the same bits
        gen = torch.Generator(device="cuda").manual_seed(321)
        batch = torch.randn((big, 3, side, side), generator=gen, device="cuda")
        at = [big // 2 + 3 + i for i in range(nb)]
        batch[at] = px
        outb = m.compute_logits(batch, [enc])
        assert outb["logits"].shape == (big, 14) and torch.isfinite(outb["logits"]).all() and torch.isfinite(outb["similarity_scores"]).all()
        b_s, br_s, b_l, simb, lgb = _errors(outb, at, g)
        hitsb = _check_argmax(simb, lgb, g, b_s)
        print(f"[{name} fp32 default, inside B = {big} (MX form)] max|dscores|={b_s:.2e} rms {br_s:.2e} max|dlogits|={b_l:.2e} patch argmax {hitsb}")
        assert b_s <= FP32_TOL and b_l <= FP32_TOL
        assert m.guard_reruns() == 0
        # one arithmetic whatever the batch: an image's bits alone and inside the timed batch are identical; the three-plane form is the other arithmetic
        assert torch.equal(outb["similarity_scores"][at], alone)
        assert not torch.equal(three, alone)
        # determinism of the timed form
        again = m.compute_logits(batch, [enc])
        assert torch.equal(again["similarity_scores"], outb["similarity_scores"]) and torch.equal(again["logits"], outb["logits"])
    finally:
        m.close()


# gates (scores max / rms, logits max) >= 1.5 x measured at these shapes on MI355X (round 6: bf16 0.050 / 0.0072 / 0.0149 at 1024^2, 0.064 / 0.0072 / 0.0163 at
# 518^2; fp16 0.0089 / 0.0011 / 0.0026 and 0.0100 / 0.0013 / 0.0027); the scores' maximum keeps G8's gate (heavy-tailed: 0.05-0.13 across equivalent roundings)
@pytest.mark.parametrize("dtype,gates", [(torch.bfloat16, {"g14_outlier_s1024_b1_t14": (0.17, 0.013, 0.026), "g15_outlier_s518_b2_t14": (0.17, 0.013, 0.026)}),
                                         (torch.float16, {"g14_outlier_s1024_b1_t14": (0.017, 0.0019, 0.0045), "g15_outlier_s518_b2_t14": (0.017, 0.0019, 0.0045)})])
def test_16bit_modes_on_the_outlier_checkpoint_at_timed_shapes(dtype, gates, cfg, outlier_sd):
    from radzero_amd.modeling import RadZeroModel
    m = RadZeroModel.from_state_dict(outlier_sd, cfg, torch_dtype=dtype, device="cuda:0").eval()
    failed = []
    try:
        for name, (s_tol, s_rms, l_tol) in gates.items():
            g = load_golden(name)
            px, enc = _golden_inputs(g)
            out = m.compute_logits(px, [enc])
            e_s, r_s, e_l, sim, lg = _errors(out, list(range(int(g["batch"]))), g)
            assert np.isfinite(sim).all() and np.isfinite(lg).all()
            print(f"\n[{name} {dtype}] max|dscores|={e_s:.5f} rms {r_s:.5f} max|dlogits|={e_l:.5f}")
            if not (e_s <= s_tol and r_s <= s_rms and e_l <= l_tol):
                failed.append((name, e_s, r_s, e_l))
            assert np.array_equal(lg.argmax(1), np.atleast_2d(g["logits"]).argmax(1))
        assert not failed, failed
    finally:
        m.close()
