"""Guards on the ISA properties the hot kernels' performance rests on (CPU only: hipcc cross-compiles gfx950 here).

The kernels are compiled device-only to assembly -- the P = 36 instantiations alone (-DE2VQ_PRE_NC_LIST / -DE2VQ_MFMA_NC_LIST),
same flags as the product build -- and checked for what a compiler release could silently take away:
no scratch, no spilled VGPR, the register budget of the occupancy the launch bounds assume, the MFMA count of the
rotating tile loop, no full vector-memory wait inside it, `s_nop 4` in front of every inline-asm operand load.
A compile with -DE2VQ_PRE_ORDER=1 (the granule-major k-step order that is known to need 38 registers too many) must
trip the guard: that is the test of the test."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ecoz2rs_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-x", "hip", "--cuda-device-only", "-S",
         "-DE2VQ_PRE_NC_LIST(X)=X(37)", "-DE2VQ_MFMA_NC_LIST(X)=X(37)", "-DE2VQ_MFMA_WIDE_NC_LIST(X)="]

pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")


def compile_asm(src, out, extra=()):
    subprocess.run([HIPCC, *FLAGS, *extra, "-o", out, os.path.join(CSRC, src)], check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL, timeout=900)
    return open(out).read()


@pytest.fixture(scope="module")
def asm(tmp_path_factory):
    d = tmp_path_factory.mktemp("isa")
    return {name: compile_asm(name + ".hip", str(d / (name + ".s"))) for name in ("vq_prefilter", "vq_device", "vq_sweep")}


class Kernel:
    def __init__(self, text, pattern):
        """pattern: regular expression matched against the mangled kernel name (exactly one kernel must match)"""
        metas = [(m.group(1), m.group(2)) for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", text, re.S)
                 if re.search(pattern, m.group(1))]
        assert len(metas) == 1, f"{pattern}: {[n for n, _ in metas]}"
        self.name, meta = metas[0]
        g = lambda k: int(re.search(r"\." + k + r":\s+(\d+)", meta).group(1))
        self.vgpr, self.spill, self.scratch = g("vgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size")
        lines = text.split("\n")
        start = next(i for i, l in enumerate(lines) if l.startswith(self.name + ":"))
        end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
        self.body = lines[start:end]

    def count(self, mnemonic):
        return sum(1 for l in self.body if l.strip().startswith(mnemonic))

    def loops(self):
        """(first line, last line, MFMAs, full vector-memory waits) of every backward branch's body"""
        labels = {m.group(1): i for i, l in enumerate(self.body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
        out = []
        for i, l in enumerate(self.body):
            m = re.match(r"\s+s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
            if m and labels.get(m.group(1), i) < i:
                seg = self.body[labels[m.group(1)]:i]
                out.append((labels[m.group(1)], i, sum("v_mfma" in x for x in seg), sum("vmcnt(0)" in x for x in seg)))
        return out

    def asm_blocks(self):
        blocks, i = [], 0
        while i < len(self.body):
            if "#ASMSTART" in self.body[i]:
                j = i
                while "#ASMEND" not in self.body[j]:
                    j += 1
                blocks.append([x.strip() for x in self.body[i + 1:j]])
                i = j
            i += 1
        return blocks

    def violations(self, max_vgpr):
        v = []
        if self.scratch != 0:
            v.append(f"{self.name}: private_segment_fixed_size {self.scratch}")
        if self.spill != 0:
            v.append(f"{self.name}: {self.spill} spilled VGPRs")
        if self.vgpr > max_vgpr:
            v.append(f"{self.name}: {self.vgpr} VGPRs > {max_vgpr}")
        return v


# (mangled-name pattern, file, VGPR budget: 512 / waves per SIMD the launch bounds assume)
CLEAN = [
    (r"k_pass_pre_ldsILi37ELb1ELi2E", "vq_prefilter", 256),   # round 4's fused pass (M = 128; ECOZ2_VQ_ACCUMULATE=records)
    (r"k_pass_pre_ldsILi37ELb1ELi1E", "vq_prefilter", 256),   # ... with the burst of atomics
    (r"k_pass_preILi37ELi6ELi512ELb1E", "vq_prefilter", 256), # fused quantize, rotating tile loop (round 6: one register set)
    (r"k_pass_preILi37ELi6ELi512ELb0E", "vq_prefilter", 256), # ... the plain loop (a single tile)
    (r"k_reduce_recordsILi37E", "vq_prefilter", 85),          # three workgroups of eight waves per CU
    (r"k_pass_mfmaILi37ELi2ELi512ELi0E", "vq_device", 256),   # plain FP64 sweep
    (r"k_pass_smallILi37E", "vq_device", 256),                # M <= 16
    (r"k_sweep_candILi37ELb1ELb1ELb0E", "vq_sweep", 256),     # round 5: the fused pass over grouped frames, two-stage sweep, two blocks per turn
    (r"k_sweep_candILi37ELb1ELb1ELb1E", "vq_sweep", 256),     # ... one block per turn (small shards)
    (r"k_sweep_candILi37ELb0ELb1ELb0E", "vq_sweep", 256),     # ... one-stage sweep
    (r"k_sweep_candILi37ELb1ELb0ELb0E", "vq_sweep", 256),     # candidate sweep alone (finishing kernel + reduce behind it)
    (r"k_sweep_candILi37ELb0ELb0ELb0E", "vq_sweep", 256),
    (r"k_finishILi37E", "vq_sweep", 168),                     # twelve waves per workgroup: three per SIMD
]


@pytest.mark.parametrize("pattern,unit,max_vgpr", CLEAN, ids=[c[0] for c in CLEAN])
def test_no_scratch_no_spill_register_budget(asm, pattern, unit, max_vgpr):
    assert Kernel(asm[unit], pattern).violations(max_vgpr) == []


@pytest.mark.parametrize("pattern", [r"k_pass_pre_ldsILi37ELb1ELi2E", r"k_pass_pre_ldsILi37ELb1ELi1E"])
def test_rotating_tile_loop(asm, pattern):
    k = Kernel(asm["vq_prefilter"], pattern)
    # tiles 0, 1 + the rotating loop's two tiles + the last two: 6 tiles x 2 jobs x 15 k-steps
    assert k.count("v_mfma_f32_32x32x16_f16") == 180
    inner = [lp for lp in k.loops() if lp[2] > 0]
    inner.sort(key=lambda lp: lp[1] - lp[0])
    first, last, mfmas, full_waits = inner[0]
    assert mfmas == 60, "the rotating loop holds two tiles of 2 x 15 MFMAs"
    assert full_waits == 0, "a vmcnt(0) inside the rotating tile loop waits for the prefetch of the tile after next"
    body = k.body[first:last]
    assert not any(l.strip().startswith(("scratch_", "buffer_load", "buffer_store")) for l in body)
    # operand loads of the loop are the inline-asm ones, each behind its hazard padding (v_readlane -> VMEM address)
    loads = [b for b in k.asm_blocks() if any("global_load_dwordx4" in x for x in b)]
    assert len(loads) == 36  # four loading tiles x 9 unique granules
    assert all(b[0].startswith("s_nop 4") for b in loads)
    assert not any(l.strip().startswith("global_load_dwordx4") for l in body
                   if not any(l.strip() in b for b in loads)), "a compiler-visible load inside the tile loop"


def test_quantize_rotating_tile_loop(asm):
    """Round 6: the fused quantize kernel's tile loop -- ONE register set, the next tile requested by inline asm behind the
    current tile's last readers, waited for granule by granule.  Its waits count the vector-memory operations in flight, so
    nothing else may issue one inside the loop: no scratch access (two register sets spilled 36 registers INTO the loop), no
    compiler-visible load, no full wait."""
    k = Kernel(asm["vq_prefilter"], r"k_pass_preILi37ELi6ELi512ELb1E")
    assert k.count("v_mfma_f32_32x32x16_f16") == 90  # tile 0, the loop's tile, the last tile: 3 x 2 jobs x 15 k-steps
    inner = [lp for lp in k.loops() if lp[2] == 30]
    assert inner, [lp for lp in k.loops() if lp[2]]
    first, last, _mf, full_waits = min(inner, key=lambda lp: lp[1] - lp[0])
    # (one full wait is the loop's own: the granule requested last is waited for with vmcnt(NU - 1 - rank) = vmcnt(0))
    assert full_waits == 1
    body = k.body[first:last]
    assert not any(l.strip().startswith(("scratch_", "buffer_load", "buffer_store")) for l in body)
    loads = [b for b in k.asm_blocks() if any("global_load_dwordx4" in x for x in b)]
    assert len(loads) == 27  # tile 0 (requested whole) + the two loading jobs x 9 unique granules
    assert all(b[0].startswith("s_nop 4") for b in loads)
    assert not any(l.strip().startswith("global_load_dwordx4") for l in body
                   if not any(l.strip() in b for b in loads)), "a compiler-visible load inside the tile loop"


@pytest.mark.parametrize("pattern,coarse_mfmas", [(r"k_sweep_candILi37ELb1ELb1ELb0E", 64), (r"k_sweep_candILi37ELb1ELb1ELb1E", 32)])
def test_two_stage_sweep_shape(asm, pattern, coarse_mfmas):
    k = Kernel(asm["vq_sweep"], pattern)
    # the coarse stage's loop: two register sets = two tiles of 4 column blocks (two blocks of 64 slots per turn: a loaded
    # tile serves four jobs) x 8 k-steps, no 15-step job inside it, and the MFMAs of a job interleaved with the previous
    # job's epilogue (never eight in a row)
    coarse = [lp for lp in k.loops() if lp[2] == coarse_mfmas]
    assert coarse, [lp for lp in k.loops() if lp[2]]
    first, last, _, _ = min(coarse, key=lambda lp: lp[1] - lp[0])
    run, longest = 0, 0
    for l in k.body[first:last]:
        t = l.strip()
        if t.startswith("v_mfma"):
            run += 1
            longest = max(longest, run)
        elif t.startswith("v_"):
            run = 0
    assert longest <= 4, "the coarse jobs' MFMAs are no longer interleaved with the key epilogue"
    assert k.count("scratch_") == 0


def test_guard_trips_on_the_known_bad_order(tmp_path):
    """-DE2VQ_PRE_ORDER=1: all three weight levels' accumulators come to life at once in a job; the rotating kernel then
    spills (docs/HISTORY.md, round 4).  The guard has to see it."""
    text = compile_asm("vq_prefilter.hip", str(tmp_path / "order1.s"), extra=["-DE2VQ_PRE_ORDER=1"])
    k = Kernel(text, r"k_pass_pre_ldsILi37ELb1ELi2E")
    assert k.violations(256) != []
