"""Can limb products of the prefiltered sweep be skipped EXACTLY?  (round 5, VERDICT task 2: measure before building)

Runs the real ladder on the bench data through the session API; at the M = 256 / 512 / 1024 levels it takes the codebook of
every pass and evaluates, with torch f64 on the GPU (diagnostics: not the product path), two schemes that would be exact by
construction:

 (a) drift bound.  d'(r, m) - d(r, m) = <r, D_m>, D_m = cq'_m - cq_m.  A frame whose gap to its runner-up exceeds
     |<r, D_m1>| + max_m |<r, D_m>| keeps its cell after the update.  What a kernel could know without a sweep:
       a1: A_t (delta_m1 + max_m delta_m),                 delta_m = sum_n a_n |D_m[n]|          (|r[n]| <= a_n A_t)
       a2: A_t g_t (linf_m1 + max_m linf_m),               linf_m  = max_n a_n |D_m[n]|, g_t = sum_n |xi_n| (stored per frame)
     and the gap it knows is the key gap minus twice the three-limb tolerance.  Reported: fraction of frames each bound
     certifies, next to the fraction that really keeps its cell (the ceiling of any such scheme).
 (b) two-stage keys.  W0 + W1 (8 of 15 k-steps) for every tile; W2 only for the (64-frame block, 32-codeword tile) pairs
     in which some frame's coarse key is within twice the two-limb tolerance of that frame's coarse minimum.  Reported: the
     fraction of (block, tile) pairs that must finish, frames in natural order and grouped by their previous cell, and
     the fraction of (frame, codeword) pairs (the ceiling at granularity 1 x 1).
Scales (a_n, A_t, C) are the kernel's (vq_prefilter.hip: k_pre_exponents, k_pre_frames, k_pre_cmax)."""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import numpy as np
import torch

import ecoz2rs_amd as e

P, NC = 36, 37
S = int(os.environ.get("SKIP_FRAMES", str(1 << 21)))
LEVELS = [int(x) for x in os.environ.get("SKIP_LEVELS", "256,512,1024").split(",")]
dev = "cuda:0"


def ref2cq(refl):
    """reflections (M, P+1) -> cq (M, P+1): step-up, autocorrelation of the predictor polynomial, doubled tail"""
    M = refl.shape[0]
    a = np.zeros((M, NC))
    a[:, 0] = 1.0
    for k in range(1, P + 1):
        akk = refl[:, k].copy()
        a[:, k] = akk
        for i in range(1, (k >> 1) + 1):
            ai, aj = a[:, i].copy(), a[:, k - i].copy()
            a[:, i] = ai + akk * aj
            a[:, k - i] = aj + akk * ai
    raa = np.zeros((M, NC))
    for n in range(NC):
        raa[:, n] = (a[:, : NC - n] * a[:, n:]).sum(1)
    cq = 2.0 * raa
    cq[:, 0] = raa[:, 0]
    return cq


def limbs(x, nl):
    """x in (-1, 1) -> its nl integer limbs X1, X2, ... (x = X1 2^-9 + X2 2^-18 + ...), as the kernel's pre_split"""
    s = x * 512.0
    parts = []
    for _ in range(nl):
        l = torch.round(s)
        parts.append(l)
        s = (s - l) * 512.0
    return parts


frames = e.synth.synth_frames(20244, 20, P, 0, S)
R = torch.from_numpy(frames).to(dev)
sess = e.VqSession(P, device=0)
sess.set_frames(frames)
del frames
sess.prepare()
sess.init_codebook()
os.environ["ECOZ2_VQ_QUIET"] = "1"

# the kernel's scales
colmax = R.abs().amax(0)
ea = torch.where(colmax > 0, torch.floor(torch.log2(colmax)) + 1, torch.zeros_like(colmax))  # a_n = 2^ea > max |r[n]|
a_n = torch.pow(2.0, ea)
Rn = R / a_n
eA = torch.floor(torch.log2(Rn.abs().amax(1).clamp_min(1e-300))) + 1
A_t = torch.pow(2.0, eA)
XI = Rn / A_t[:, None]          # in (-1, 1)
g_t = XI.abs().sum(1)
X = limbs(XI, 3)                # three integer limb matrices

CH = 1 << 16


def sweep(cq):
    """exact-ish f64 distances: (d1, m1, d2) per frame"""
    C = torch.from_numpy(cq).to(dev)
    d1 = torch.empty(S, dtype=torch.float64, device=dev)
    d2 = torch.empty_like(d1)
    m1 = torch.empty(S, dtype=torch.int64, device=dev)
    for o in range(0, S, CH):
        D = R[o:o + CH] @ C.T
        v, i = torch.topk(D, 2, dim=1, largest=False)
        d1[o:o + CH], d2[o:o + CH], m1[o:o + CH] = v[:, 0], v[:, 1], i[:, 0]
    return d1, m1, d2


def codebook_scales(cq):
    C = torch.from_numpy(cq).to(dev)
    Ca = C * a_n
    eC = torch.floor(torch.log2(Ca.abs().amax())) + 1
    Cs = torch.pow(2.0, eC)
    ETA = Ca / Cs
    return C, Cs, ETA


def report_b(cq, groupings, tag):
    """groupings: list of (name, per-frame sort key, per-frame home codeword or None)"""
    C, Cs, ETA = codebook_scales(cq)
    M = C.shape[0]
    MT = M // 32
    Y = limbs(ETA, 3)
    ymax = ETA.abs().sum(1).amax()
    # coarse key = 2^-18 W0 + 2^-27 W1 (units of sum xi eta); rigorous two-limb tolerance:
    #   |W2| 2^-36 <= 2^-19 (g + ymax) + NC 2^-20 (+ the three-limb remainder, + f32 rounding of the key)
    tol2 = (2.0 ** -19) * 1.002 * (g_t + ymax) + NC * 2.0 ** -20 + 2.0 ** -22
    delta = 2.0 * 1.27 * tol2
    need_pairs = 0
    nat_blocks = torch.zeros((S // 64, MT), dtype=torch.bool, device=dev)
    grp = []
    for name, key_, home in groupings:
        order = torch.argsort(key_, stable=True)
        slot_of = torch.empty(S, dtype=torch.int64, device=dev)
        slot_of[order] = torch.arange(S, device=dev)
        # home tile of a block = tile of the home codeword of the block's first frame
        home_tile = (home[order[::64]] // 32) if home is not None else None
        grp.append([name, slot_of // 64, home_tile, torch.zeros((S // 64, MT), dtype=torch.bool, device=dev),
                    torch.zeros((S // 64, MT), dtype=torch.bool, device=dev)])
    Y0, Y1 = Y[0].T.contiguous(), Y[1].T.contiguous()
    ti = torch.arange(MT, device=dev)
    for o in range(0, S, CH):
        x0, x1 = X[0][o:o + CH], X[1][o:o + CH]
        W0 = x0 @ Y0
        W1 = x0 @ Y1 + x1 @ Y0
        key = W0 * 2.0 ** -18 + W1 * 2.0 ** -27
        kmin = key.amin(1, keepdim=True)
        rel = 2.0 ** -22 * key.abs()
        need = key <= kmin + delta[o:o + CH, None] + rel
        need_pairs += int(need.sum())
        nt = need.view(-1, MT, 32).any(2)                      # (frame, tile)
        nat_blocks[o // 64:(o + CH) // 64] = nt.view(-1, 64, MT).any(1)
        tmin = key.view(-1, MT, 32).amin(2)                    # (frame, tile): smallest coarse key of the tile
        for g in grp:
            gb = g[1][o:o + CH]
            gbx = gb[:, None].expand(-1, MT)
            tix = ti[None, :].expand(nt.shape[0], -1)
            g[3][gbx[nt], tix[nt]] = True
            if g[2] is not None:  # pessimistic: U = the frame's smallest coarse key in its block's home tile, never tightened
                U = tmin.gather(1, g[2][gb][:, None])
                nh = tmin <= U + delta[o:o + CH, None] + 2.0 ** -22 * tmin.abs()
                g[4][gbx[nh], tix[nh]] = True
    msg = (f"  (b) {tag}: median two-limb tolerance {float(tol2.median()):.3e}; (frame, codeword) pairs within reach "
           f"{need_pairs / (S * M):.3f}; (block, tile) pairs that must finish W2: natural order {float(nat_blocks.float().mean()):.3f}")
    for g in grp:
        f_opt = float(g[3].float().mean())
        msg += f"; grouped by {g[0]}: {f_opt:.3f} (k-steps {8 / 15 + f_opt:.3f} of today's when a flagged tile reruns all 15)"
        if g[2] is not None:
            f_home = float(g[4].float().mean())
            msg += f", with U fixed from the home tile {f_home:.3f} ({8 / 15 + f_home:.3f})"
    print(msg, flush=True)


def report_a(cq_old, cq_new, d1, m1, d2, tag):
    Co, Cso, ETAo = codebook_scales(cq_old)
    Cn = torch.from_numpy(cq_new).to(dev)
    D = (Cn - Co)
    Da = D.abs() * a_n
    delta = Da.sum(1)
    linf = Da.amax(1)
    ymax = ETAo.abs().sum(1).amax()
    # three-limb key tolerance in units of d: A_t C 2^-36 * 1.27 * 2^8 (g + ymax + NC + 4)   (the kernel's tau without the relative term)
    tol3 = A_t * Cso * 2.0 ** -36 * 1.27 * 256.0 * (g_t + ymax + NC + 4.0)
    gap = (d2 - d1) - 2.0 * tol3
    b1 = A_t * (delta[m1] + delta.amax())
    b2 = A_t * g_t * (linf[m1] + linf.amax())
    bb = torch.minimum(b1, b2)
    # ceiling: frames that really keep their cell
    _, m1n, _ = sweep(cq_new)
    stay = (m1n == m1).float().mean()
    # the exact drift (a sweep's worth of work: only as a yardstick): max_m <r, D_m> - <r, D_m1>
    ex_ok = 0
    for o in range(0, S, CH):
        dr = R[o:o + CH] @ D.T
        own = dr.gather(1, m1[o:o + CH, None])[:, 0]
        ex_ok += int(((own - dr.amin(1)) < gap[o:o + CH]).sum())
    print(f"  (a) {tag}: frames that keep their cell {float(stay):.3f}; certified by a1 {float((gap > b1).float().mean()):.4f}, by a2 "
          f"{float((gap > b2).float().mean()):.4f}, by min {float((gap > bb).float().mean()):.4f}; by the EXACT drift (costs a sweep) "
          f"{ex_ok / S:.3f}; median gap {float(gap.median()):.3e} vs median bound {float(bb.median()):.3e} "
          f"(rel. codeword change max_m delta_m / |cq|_1: {float((delta / (Co.abs() * a_n).sum(1)).amax()):.2e})", flush=True)


prev_cells = None
m = 2
top = max(LEVELS)
while m <= top:
    if m not in LEVELS:
        sess.learn(0.05, m)
        m *= 2
        continue
    assert sess.codebook_size() == m // 2
    # cells at the end of the previous level (the grouping a sorted layout would use on the first pass)
    cq_prev = ref2cq(sess.get_codebook())
    _, prev_cell, _ = sweep(cq_prev)
    sess.grow()
    print(f"level M = {m}", flush=True)
    dd_prev = sess.prev_distortion()
    p = 0
    while True:
        cq = ref2cq(sess.get_codebook())
        d1, m1, d2 = sweep(cq)
        gl = [("cell at level start", prev_cell, 2 * prev_cell)]
        if p > 0:
            gl.append(("cell of the previous pass", cell_last, cell_last))
        report_b(cq, gl, f"pass {p + 1}")
        sess.run_pass()
        st = sess.pass_stats()
        cell_last = m1
        done = p > 0 and (dd_prev - st.DD) / st.DD < 0.05
        dd_prev = st.DD
        if done:
            break
        sess.update()
        cq_new = ref2cq(sess.get_codebook())
        report_a(cq, cq_new, d1, m1, d2, f"pass {p + 1} -> {p + 2}")
        p += 1
    sess.set_prev_distortion(dd_prev)
    m *= 2
print("done", flush=True)
