"""Generate tools/experiments/attn_ks_loop.inc: the hot loop of the key-split flash-attention kernel (attention.hip,
flash_attn_ks_kernel) as inline-asm text.  Run:  python tools/gen_attn_ks_loop.py   (writes the .inc; commit it).

Why generated: one wave per SIMD owns 128 query rows and half of every 64-key tile (~450 live registers), so nothing but the wave's own
instruction order can overlap the exponentials with the MFMAs — the loop is software-pipelined by hand and hipcc's allocator cannot hold
the register map (the C++ form of the same kernel spills 70-90 registers and runs 40 % slower than the 4 x 32-row kernel).

Register map (asm-owned):
  a[0:127]    O^T accumulators  oacc[a][dt] at 4 (4 a + dt)        (a = 16-row query tile 0..7, dt = 16-row tile of d)      OUTPUT
  a[128:159]  row sums          lacc[a]     at 128 + 4 a                                                                     OUTPUT
  a[160:223]  Q fragments       qf[a][ks]   at 160 + 4 (2 a + ks)   (B operand of the score MFMAs, read straight from the a-file; loaded here)
  v[0:63], v[64:127]  score sets 0 / 1: sacc[a][kt] at 64 set + 4 (2 a + kt)   (tile t lives in set t & 1)
  v[128:159]  P^T fragments     pf[a] at 128 + 4 a
  v[160:175]  K fragments       kf[ks][kt] at 160 + 4 (2 ks + kt)
  v[176:191]  V^T fragments     vf[dt] at 176 + 4 dt
  v[192:199]  C operand of the first score MFMA of key tile kt (0, 1): zeros, except for the ragged last tile where the registers of
              keys >= n_valid hold -inf (the mask costs no instruction in the loop).  No reference point is subtracted: P = 2^S with S in
              log2 units as it comes — exact in floating point as long as no 2^S leaves the f32 range; the caller checks l and O and
              falls back to the tracked-maximum path otherwise.
  v[200:203]  all-ones A fragment;  v[204:206] LDS read addresses (K ks 0, K ks 1, V^T);  v207 zero, v208 -inf
LDS: five 16 KB tile buffers (K 8 KB, then V^T 8 KB), tile s in buffer s % 5.

Pipeline, iteration t (tile t's scores S(t) were produced by iteration t-1 into set t & 1):
  X_t   s_waitcnt vmcnt(4); s_barrier        tile t+2 has landed everywhere; every wave is done with tile t-1's buffer
        the wave's 4 LDS-DMA pieces of tile t+4 (into tile t-1's buffer), spread over P3
  P3    20 MFMAs: O^T, l += V^T(t-1) P^T(t-1) for query tiles 4..7                      (not in iteration 0)
        4 ds_read_b128: V^T(t) -> vf
  P1    32 MFMAs: S(t+1) = K(t+1) Q^T - m into the OTHER score set   ||   VALU: P(t) = 2^S(t), packed, for query tiles 0..4
        4 ds_read_b128: K(t+2) -> kf
  P2    20 MFMAs: O^T, l += V^T(t) P^T(t) for query tiles 0..3       ||   VALU: the same for query tiles 5..7
after the last iteration: its P3.  Tiles past the last one are staged clamped to it and their scores are never used.
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "experiments", "attn_ks_loop.inc")

NBUF, BUF = 5, 16384
S_K, S_V, S_T, S_TS, S_DB, S_KB, S_VB, S_CNT, S_IK, S_IV = 64, 66, 68, 70, 71, 72, 73, 74, 75, 76
V_ONES, V_KA0, V_KA1, V_VA, V_ZERO = 200, 204, 205, 206, 207
S_Q, S_NEXT, S_MK = 78, 77, 80


def oacc(a, dt): return f"a[{4 * (4 * a + dt)}:{4 * (4 * a + dt) + 3}]"
def lacc(a): return f"a[{128 + 4 * a}:{128 + 4 * a + 3}]"
def qf(a, ks): return f"a[{160 + 4 * (2 * a + ks)}:{160 + 4 * (2 * a + ks) + 3}]"
def sacc(st, a, kt): return f"v[{64 * st + 4 * (2 * a + kt)}:{64 * st + 4 * (2 * a + kt) + 3}]"
def sreg(st, a, kt, r): return f"v{64 * st + 4 * (2 * a + kt) + r}"
def pf(a): return f"v[{128 + 4 * a}:{128 + 4 * a + 3}]"
def kf(ks, kt): return f"v[{160 + 4 * (2 * ks + kt)}:{160 + 4 * (2 * ks + kt) + 3}]"
def vf(dt): return f"v[{176 + 4 * dt}:{176 + 4 * dt + 3}]"
def cmask(kt): return f"v[{192 + 4 * kt}:{192 + 4 * kt + 3}]"


ABL = set()          # timing ablations of the generated text (WRONG results): novalu, nodma, nords, nobar, nop1mfma, nopvmfma


class Emit:
    def __init__(self): self.lines = []
    def __call__(self, s):
        if 'nodma' in ABL and s.startswith("global_load_lds"): return
        if 'nords' in ABL and s.startswith("ds_read"): return
        if 'nobar' in ABL and s.startswith("s_barrier"): return
        if 'novalu' in ABL and (s.startswith("v_exp") or s.startswith("v_cvt_pk")) and "v215" not in s: return
        if 'expmov' in ABL and s.startswith("v_exp_f32"): s = s.replace("v_exp_f32", "v_mov_b32")
        if 'nocvt' in ABL and s.startswith("v_cvt_pk"): return
        if 'expsrc' in ABL and s.startswith("v_exp_f32 v") and not s.startswith("v_exp_f32 v2"): s = "v_exp_f32 v209, v215"
        if 'cvtsrc' in ABL and s.startswith("v_cvt_pk") and "v215" not in s: s = s.split(",")[0].split()[0] + " v210, v215, v214"
        if 'cvtdst' in ABL and s.startswith("v_cvt_pk") and "v215" not in s: s = s.split()[0] + " v210," + s.split(",", 1)[1]
        self.lines.append(s)


def mfma(e, mn, d, a, b, c):
    e(".p2align 3")                    # code placement: see tools/gen_gemm10_kloop.py
    e(f"v_mfma_f32_16x16x32_{mn} {d}, {a}, {b}, {c}")
    for t in ABL:                      # probe: k independent full-rate VALU instructions behind every MFMA
        if t.startswith("mov"):
            for i in range(int(t[3:])): e(f"v_mov_b32 v{209 + i}, v215")
        if t.startswith("fexp"):
            for i in range(int(t[4:])): e(f"v_exp_f32 v{209 + i}, v215")
        if t.startswith("fcvt"):
            for i in range(int(t[4:])): e(f"v_cvt_pk_bf16_f32 v{209 + i}, v215, v214")


def rotate(e, s):
    e(f"s_add_u32 s{s}, s{s}, {BUF}")
    e(f"s_cmp_eq_u32 s{s}, {NBUF * BUF}")
    e(f"s_cselect_b32 s{s}, 0, s{s}")


def dma_piece(p):
    """piece p of the wave's four per tile: K rows (p = 0, 1), V^T rows (p = 2, 3); destination buffer at s[S_DB]"""
    name, src = (("ldsk", "dk"), ("ldsv", "dv"))[p >> 1], (S_K, S_V)[p >> 1]
    return [f"s_add_u32 m0, s{S_DB}, %[{name[0]}{p & 1}]", "s_nop 0", f"global_load_lds_dwordx4 %[{name[1]}{p & 1}], s[{src}:{src + 1}]"]


def dma_advance():
    """after a tile's four pieces: sources move one tile on unless that was the last tile (clamp), destination to the next buffer"""
    return [f"s_add_u32 s{S_TS}, s{S_TS}, 1", f"s_cmp_lt_u32 s{S_TS}, %[ntl]",
            f"s_cselect_b32 s{S_IK}, 8192, 0", f"s_cselect_b32 s{S_IV}, 128, 0",
            f"s_add_u32 s{S_K}, s{S_K}, s{S_IK}", f"s_addc_u32 s{S_K + 1}, s{S_K + 1}, 0",
            f"s_add_u32 s{S_V}, s{S_V}, s{S_IV}", f"s_addc_u32 s{S_V + 1}, s{S_V + 1}, 0",
            f"s_add_u32 s{S_DB}, s{S_DB}, {BUF}", f"s_cmp_eq_u32 s{S_DB}, {NBUF * BUF}", f"s_cselect_b32 s{S_DB}, 0, s{S_DB}"]


def stage_tile(e):
    for p in range(4):
        for ln in dma_piece(p): e(ln)
    for ln in dma_advance(): e(ln)


def softmax_valu(cvt, st):
    """P = 2^S in place and the four packed pairs per query tile (pack8's order: key tile kt = 0 rows 0..3, then kt = 1), for all eight
    query tiles: 96 instructions.  The packs of query tile a come AFTER the exponentials of tile a+1 — the wave is alone on its SIMD, an
    instruction that waits for a transcendental result stalls everything behind it, the next MFMA included."""
    def exps(a): return [f"v_exp_f32 {sreg(st, a, kt, r)}, {sreg(st, a, kt, r)}" for kt in range(2) for r in range(4)]
    def packs(a): return [f"{cvt} v{128 + 4 * a + i}, {sreg(st, a, i >> 1, 2 * (i & 1))}, {sreg(st, a, i >> 1, 2 * (i & 1) + 1)}" for i in range(4)]
    out = exps(0)
    for a in range(1, 8): out += exps(a) + packs(a - 1)
    return out + packs(7)


def interleave(e, mn, mfmas, others):
    """issue the MFMAs in order with the other instructions spread evenly between them"""
    n, m = len(mfmas), len(others)
    if n == 0:
        for ln in others: e(ln)
        return
    done = 0
    for i, (d, a, b, c) in enumerate(mfmas):
        mfma(e, mn, d, a, b, c)
        upto = (i + 1) * m // n
        for ln in others[done:upto]: e(ln)
        done = upto


def pv_mfmas(group):
    """O^T += V^T P^T and l += 1 P^T for the query tiles of `group`"""
    out = [(oacc(a, dt), vf(dt), pf(a), oacc(a, dt)) for dt in range(4) for a in group]
    out += [(lacc(a), f"v[{V_ONES}:{V_ONES + 3}]", pf(a), lacc(a)) for a in group]
    return out


def score_mfmas(st):
    out = [(sacc(st, a, kt), kf(0, kt), qf(a, 0), cmask(kt)) for kt in range(2) for a in range(8)]
    out += [(sacc(st, a, kt), kf(1, kt), qf(a, 1), sacc(st, a, kt)) for kt in range(2) for a in range(8)]
    return out


def mask_check(e):
    """before the scores of tile s[S_NEXT] are computed: the ragged last tile gets its -inf C operand (lane's key j = 4 kt + r of its 8-key
    group is dead when j >= %[vg]); every other tile keeps zeros"""
    e(f"s_cmp_eq_u32 s{S_NEXT}, s{S_MK}")
    e("s_cbranch_scc0 9f")
    for j in range(8):
        e(f"v_cmp_lt_i32 vcc, {j}, %[vg]")
        e(f"v_cndmask_b32 v{192 + j}, v{V_ZERO + 1}, v{V_ZERO}, vcc")
    e("9:")
    e(f"s_add_u32 s{S_NEXT}, s{S_NEXT}, 1")


def k_reads(e):
    e(f"v_add_u32 v{V_KA0}, s{S_KB}, %[koff0]")
    e(f"v_add_u32 v{V_KA1}, s{S_KB}, %[koff1]")
    for ks in range(2):
        for kt in range(2): e(f"ds_read_b128 {kf(ks, kt)}, v{V_KA0 + ks} offset:{kt * 512}")
    rotate(e, S_KB)


def v_reads(e):
    e(f"v_add_u32 v{V_VA}, s{S_VB}, %[voff]")
    for dt in range(4): e(f"ds_read_b128 {vf(dt)}, v{V_VA} offset:{dt * 2048}")
    rotate(e, S_VB)


def softmax_lists(cvt, cur):
    exps = [f"v_exp_f32 {sreg(cur, a, kt, r)}, {sreg(cur, a, kt, r)}" for a in range(8) for kt in range(2) for r in range(4)]
    packs = [f"{cvt} v{128 + 4 * a + i}, {sreg(cur, a, i >> 1, 2 * (i & 1))}, {sreg(cur, a, i >> 1, 2 * (i & 1) + 1)}" for a in range(8) for i in range(4)]
    return exps, packs


def emit_phase(e, mn, mfmas, per_slot):
    """MFMAs in order, slot k followed by the instructions of per_slot[k]"""
    for k, (d, a, b, c) in enumerate(mfmas):
        mfma(e, mn, d, a, b, c)
        for ln in per_slot.get(k, []):
            if 'dummyvalu' in ABL and ln.startswith("v_exp_f32"): ln = "v_exp_f32 v209, v215"
            if 'dummyvalu' in ABL and ln.startswith("v_cvt_pk"): ln = ln.split()[0] + " v210, v215, v214"
            e(ln)


def lines_of(fn):
    x = Emit(); fn(x); return x.lines


def iteration(e, mn, cvt, cur, with_p3):
    """Measured on this loop (RZ_KS_PROBE builds; one wave per SIMD, so every stall is paid in full): behind an MFMA two v_exp_f32, or one
    plain VALU instruction, are free; but a transcendental FOLLOWED by any other vector instruction (a pack, an address add, an LDS or
    LDS-DMA issue) stalls the stream for tens of cycles — one exponential and one pack per MFMA cost 45 %.  So the 64 exponentials of
    tile t go out as one run (two behind each of the first 32 MFMAs of the iteration), and everything else — packs, the V^T reads, the
    LDS-DMA issue — behind the later MFMAs."""
    exps, packs = softmax_lists(cvt, cur)
    e("s_waitcnt vmcnt(4)")
    e("s_barrier")
    dma = [ln for p in range(4) for ln in dma_piece(p)] + dma_advance()
    vr = lines_of(v_reads)
    if with_p3:
        emit_phase(e, mn, pv_mfmas(range(4, 8)), {k: exps[2 * k:2 * k + 2] for k in range(20)})
        mask_check(e)
        e("s_waitcnt lgkmcnt(0)")                    # the K fragments of tile t+1
        slots = {k: exps[40 + 2 * k:42 + 2 * k] for k in range(12)}
        slots[12] = vr + packs[0:1]
        for k in range(13, 32): slots[k] = packs[k - 12:k - 11]          # packs 1..19 (query tiles 0..3 are packs 0..15)
        for n, ln in enumerate(dma):                                     # the LDS-DMA issue behind slots 14..31
            slots[14 + n * 18 // len(dma)] = slots[14 + n * 18 // len(dma)] + [ln]
        emit_phase(e, mn, score_mfmas(cur ^ 1), slots)
        k_reads(e)
        e("s_waitcnt lgkmcnt(4)")                    # V^T(t) (older than the four K reads just issued)
        emit_phase(e, mn, pv_mfmas(range(0, 4)), {k: packs[20 + k:21 + k] for k in range(12)})
    else:
        for ln in dma: e(ln)
        for ln in vr: e(ln)
        mask_check(e)
        e("s_waitcnt lgkmcnt(4)")
        emit_phase(e, mn, score_mfmas(cur ^ 1), {k: exps[2 * k:2 * k + 2] for k in range(32)})
        for ln in packs[0:16]: e(ln)
        k_reads(e)
        e("s_waitcnt lgkmcnt(4)")
        emit_phase(e, mn, pv_mfmas(range(0, 4)), {k: packs[16 + k:17 + k] for k in range(16)})


def merge(a, b):
    """two instruction lists, each in its own order, spread evenly through one another"""
    out, ia, ib = [], 0, 0
    while ia < len(a) or ib < len(b):
        if ib >= len(b) or (ia < len(a) and ia * len(b) <= ib * len(a)):
            out.append(a[ia]); ia += 1
        else:
            out.append(b[ib]); ib += 1
    return out


def block(mn):
    cvt = {"bf16": "v_cvt_pk_bf16_f32"}[mn]
    e = Emit()
    # Q fragments straight into the a-file (16 loads per lane), then the first four tiles' LDS-DMA: one latency, not two
    e(f"s_mov_b32 s{S_Q}, %[qlo]"); e(f"s_mov_b32 s{S_Q + 1}, %[qhi]")
    for a in range(8):
        for ks in range(2): e(f"global_load_dwordx4 {qf(a, ks)}, %[qoff], s[{S_Q}:{S_Q + 1}] offset:{(a & 1) * 2048 + ks * 64}")
        if a & 1:
            e(f"s_add_u32 s{S_Q}, s{S_Q}, 4096"); e(f"s_addc_u32 s{S_Q + 1}, s{S_Q + 1}, 0")
    e(f"s_mov_b32 s{S_K}, %[klo]"); e(f"s_mov_b32 s{S_K + 1}, %[khi]")
    e(f"s_mov_b32 s{S_V}, %[vlo]"); e(f"s_mov_b32 s{S_V + 1}, %[vhi]")
    e(f"s_mov_b32 s{S_TS}, 0"); e(f"s_mov_b32 s{S_DB}, 0"); e(f"s_mov_b32 s{S_KB}, 0"); e(f"s_mov_b32 s{S_VB}, 0")
    for _ in range(4): stage_tile(e)
    for i in range(160): e(f"v_accvgpr_write_b32 a{i}, 0")
    for i in range(4): e(f"v_mov_b32 v{V_ONES + i}, 0x3f803f80")
    for i in range(8): e(f"v_mov_b32 v{192 + i}, 0")
    e(f"v_mov_b32 v{V_ZERO}, 0")
    e(f"v_mov_b32 v{V_ZERO + 1}, 0xff800000")
    e(f"s_mov_b32 s{S_NEXT}, 0")
    e(f"s_mov_b32 s{S_MK}, %[mk]")                   # index of the ragged tile (the last one), or -1
    e("s_waitcnt vmcnt(8)")                          # Q, tiles 0 and 1
    e("s_barrier")
    k_reads(e)                                       # K(0)
    mask_check(e)
    e("s_waitcnt lgkmcnt(0)")
    for m in score_mfmas(0): mfma(e, mn, *m)         # S(0)
    k_reads(e)                                       # K(1)
    # iteration 0, then pairs, then a single one if the count is even, then the last P3
    e(f"s_sub_u32 s{S_CNT}, %[n], 1")
    iteration(e, mn, cvt, 0, False)
    e(f"s_cmp_lt_u32 s{S_CNT}, 2")
    e("s_cbranch_scc1 4f")
    e("3:")
    iteration(e, mn, cvt, 1, True)
    iteration(e, mn, cvt, 0, True)
    e(f"s_sub_u32 s{S_CNT}, s{S_CNT}, 2")
    e(f"s_cmp_lt_u32 s{S_CNT}, 2")
    e("s_cbranch_scc0 3b")
    e("4:")
    e(f"s_cmp_eq_u32 s{S_CNT}, 0")
    e("s_cbranch_scc1 5f")
    iteration(e, mn, cvt, 1, True)
    e("5:")
    for m in pv_mfmas(range(4, 8)): mfma(e, mn, *m)
    e("s_waitcnt vmcnt(0) lgkmcnt(0)")
    e("s_nop 15")
    e("s_nop 15")                                    # MFMA results -> the compiler's v_accvgpr_read (it sees no MFMA here)
    return e.lines


def c_string(lines):
    return "\n".join('    "' + part + '\\n"' for ln in lines for part in ln.split("\n"))


def main():
    parts = ["// GENERATED by tools/gen_attn_ks_loop.py — do not edit.  Inline-asm text of flash_attn_ks_kernel's hot loop.\n"]
    global ABL
    variants = [("", set())]
    if os.environ.get("RZ_KS_ABLATIONS") == "1":
        variants += [("_NOVALU", {"novalu"}), ("_NODMA", {"nodma"}), ("_NORDS", {"nords"}), ("_NOBAR", {"nobar"}), ("_NOP1MFMA", {"expmov"}),
                     ("_NOPVMFMA", {"nocvt"}), ("_MFMAONLY", {"novalu", "nodma", "nords", "nobar"})]
        mo = {"novalu", "nodma", "nords", "nobar"}
        if os.environ.get("RZ_KS_PROBE") == "4":
            variants[1:] = [("_NOVALU", {"novalu"}), ("_NODMA", {"dummyvalu"}), ("_NORDS", {"dummyvalu", "nodma", "nords", "nobar"}),
                            ("_NOBAR", {"nodma", "nords", "nobar"}), ("_NOP1MFMA", {"nodma"}), ("_NOPVMFMA", {"nords"}), ("_MFMAONLY", mo)]
        if os.environ.get("RZ_KS_PROBE") == "3":
            nv = {"novalu"}
            variants[1:] = [("_NOVALU", mo | {"fexp1", "fcvt1"}), ("_NODMA", mo | {"fexp1", "mov1"}), ("_NORDS", nv), ("_NOBAR", nv | {"fexp1"}),
                            ("_NOP1MFMA", nv | {"fcvt1"}), ("_NOPVMFMA", nv | {"fexp1", "fcvt1"}), ("_MFMAONLY", mo)]
        if os.environ.get("RZ_KS_PROBE") == "2":
            variants[1:] = [("_NOVALU", {"novalu"}), ("_NODMA", {"expsrc"}), ("_NORDS", {"cvtsrc"}), ("_NOBAR", {"cvtdst"}),
                            ("_NOP1MFMA", {"expsrc", "cvtsrc"}), ("_NOPVMFMA", {"nocvt"}), ("_MFMAONLY", mo)]
        variants[1:] = [("_NOVALU", mo | {"mov1"}), ("_NODMA", mo | {"mov2"}), ("_NORDS", mo | {"mov3"}), ("_NOBAR", mo | {"fexp1"}),
                        ("_NOP1MFMA", mo | {"fexp2"}), ("_NOPVMFMA", mo | {"fcvt1"}), ("_MFMAONLY", mo)] if os.environ.get("RZ_KS_PROBE") == "1" else variants[1:]
    for mn in ("bf16",):
        for suffix, abl in variants:
            ABL = abl
            lines = block(mn)
            n_mfma = sum(1 for l in lines if l.startswith("v_mfma"))
            parts.append(f"// RZ_ATTN_KS_LOOP_{mn.upper()}{suffix}: {len(lines)} lines, {n_mfma} MFMAs in the text\n#define RZ_ATTN_KS_LOOP_{mn.upper()}{suffix} \\\n" +
                         " \\\n".join(c_string(lines).split("\n")) + "\n")
    clob = ['"memory"', '"scc"', '"vcc"', '"m0"'] + [f'"s{i}"' for i in range(64, 81)] + [f'"v{i}"' for i in range(0, 216)] + [f'"a{i}"' for i in range(160, 224)]
    parts.append("// registers the loop owns besides its pinned operands\n#define RZ_ATTN_KS_CLOBBERS " + ", ".join(clob) + "\n")
    open(OUT, "w").write("\n".join(parts))
    print("wrote", OUT, sum(len(p) for p in parts), "bytes")


if __name__ == "__main__":
    main()
