"""Guards on the machine code of the stage-2 kernels (CPU only: hipcc cross-compiles gfx950 without a GPU).

Round 2 found, in the ISA, that a prefetch written as `if (more) fetch(next); ... if (more) touch(next);` does not prefetch: the
compiler cannot pair a wait with loads that sit behind a branch and puts `s_waitcnt vmcnt(0)` in front of the FIRST USE of a group --
right after the next group's loads were issued -- so every other group paid a full memory round trip (DESIGN.md 4.2,
profiles/r02_experiments.md row O; 0.65 ms of stage 2).  Nothing in the results shows such a regression (parity is unaffected), so
the structure is pinned here: inside the hot loop of every lane-per-channel kernel the only full waits are the ones the source
asks for with touch() -- one per group of samples.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "rtlsdr-airband_amd", "csrc", "demod.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _product_flags(name):
    """The flags rtlsdr-airband_amd/_build.py compiles csrc/<name> with (optimisation level, contraction, and the library-wide device flags)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("airband_build_flags", os.path.join(ROOT, "rtlsdr-airband_amd", "_build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return list(b.HIP_SOURCES[name]) + list(b.DEVICE_FLAGS)


@pytest.fixture(scope="module")
def demod_asm(tmp_path_factory):
    if not (os.path.exists(HIPCC) or shutil.which("hipcc")):
        pytest.skip("no hipcc")
    out = str(tmp_path_factory.mktemp("isa") / "demod.s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-std=c++17"] + _product_flags("demod.hip") + ["-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", out, SRC]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL, timeout=600)
    return open(out).read().split("\n")


def _function(lines, key):
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and key in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return [l for l in lines[start:end] if not re.match(r"^\s*;", l)]


def _outer_loops(body):
    """(first index, back-edge index) of every depth-1 loop: from its header -- or from its latch, when the compiler has rotated the loop so
    that the latch block sits in FRONT of the header and falls through into it -- to the last branch that jumps back to either.  Blocks
    placed behind that branch (the seldom-taken paths the source marks with __builtin_expect) belong to the loop but not to its hot part."""
    loops = []
    label_at = [i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)]
    for n, i in enumerate(label_at):
        m = re.match(r"^(\.L(BB\d+_\d+)):.*Loop Header: Depth=1", body[i])
        if not m:
            continue
        targets, first = [m.group(1)], i
        if n > 0 and re.search(r"in Loop: Header=" + m.group(2) + r"\b", body[label_at[n - 1]]):
            prev = label_at[n - 1]
            if not any(re.match(r"^\s*(s_branch|s_endpgm|s_setpc)", l) for l in body[prev:i][-1:]):  # falls through into the header: the latch
                targets.append(body[prev].split(":")[0])
                first = prev
        back = [k for k, l in enumerate(body) if k > i and any(re.match(r"^\s*s_c?branch\w*\s+" + re.escape(t) + r"\s*$", l) for t in targets)]
        if back:
            loops.append((first, max(back)))
    return loops


def _hot_loop(body, needs):
    """The depth-1 loop with the most instructions that contains `needs` (a regex: the loop's prefetch loads)."""
    best = None
    for head, back in _outer_loops(body):
        seg = body[head:back + 1]
        if sum(1 for l in seg if re.search(needs, l)) >= 2 and (best is None or back - head > best[1] - best[0]):
            best = (head, back)
    assert best is not None, "hot loop not found"
    return body[best[0]:best[1] + 1]


@pytest.mark.parametrize("kernel,per_iteration", [
    ("demod_kernelILi0ELb0ELi1E", 2),   # AM: two groups of four samples per iteration
    ("demod_kernelILi2ELb0ELi1E", 2),   # NFM + lowpass
    ("demod_kernelILi3ELb1ELi1E", 2),   # CTCSS front
])
def test_group_prefetch_is_waited_for_once_per_group(demod_asm, kernel, per_iteration):
    loop = _hot_loop(_function(demod_asm, kernel), r"global_load_dwordx4")
    full_waits = [l for l in loop if re.search(r"s_waitcnt\s+vmcnt\(0\)", l)]
    # the rarely taken paths (AM fade-out, AGC bootstrap) sit out of line behind the loop thanks to their __builtin_expect hints;
    # should a compiler place one inside again it brings its own load + wait along: allow one, not one per sample
    assert per_iteration <= len(full_waits) <= per_iteration + 1, (
        "%s: %d full vector-memory waits inside the hot loop, %d groups per iteration -- a wait in front of a group's first use means "
        "the prefetch of the NEXT group is being waited for too (keep fetch()/touch() unconditional)" % (kernel, len(full_waits), per_iteration))
    assert sum(1 for l in loop if "global_load_dwordx4" in l) >= 2 * per_iteration


@pytest.mark.parametrize("kernel,load", [("tone_kernelILb1E", r"global_load_dword\s"), ("tone_kernelILb0E", r"global_load_dwordx2\s")],
                         ids=["one_word_hand_off", "pair_hand_off"])
def test_tone_kernel_waits_once_per_group_of_steps(demod_asm, kernel, load):
    loop = _hot_loop(_function(demod_asm, kernel), load)
    loads = [i for i, l in enumerate(loop) if re.search(load, l)]
    waits = [i for i, l in enumerate(loop) if re.search(r"s_waitcnt\s+vmcnt\(0\)", l)]
    assert len(loads) == 10, "ten steps in flight"
    assert loads[-1] < 0.2 * len(loop), "the next group's loads go out at the top of a group"
    # the wait belongs to the group boundary: at the top, in front of the loads, and / or at the bottom where the registers rotate --
    # never between the loads and the ten steps of work they are meant to fly under
    assert 1 <= len(waits) <= 2 and all(w < loads[0] or w > 0.8 * len(loop) for w in waits), waits


def test_tone_kernel_keeps_its_scalars_in_registers(demod_asm):
    """Round 2 dropped a fully unrolled variant of the tone kernel's steady-state loops after ONE failing run of the bit-exact stage-2 test;
    round 3 rebuilt that variant (50 constant-lane v_readlane in a row: 1 150 v_readlane, 38 spilled SGPRs against none) and could not make
    it fail (profiles/r03_experiments.md: 48 of 48 runs bit-exact, and the 65 536-dongle whole-handle replica test), so no hazard was found in
    the kernel.  Round 5: the steady-state recurrences take their samples from LDS broadcasts (ds_read_b128, four samples each) instead of a
    v_readlane per sample; the kernel then parks about a dozen scalars in lanes of a vector register (v_writelane / v_readlane pairs around the
    prologue, the epilogue and the rare window ends) -- validated on the GPU with that allocation (profiles/r05_tone, the parity fuzz, the whole-handle
    replica tests of the GPU suite).  What stays pinned: nothing goes to SCRATCH memory, no vector register is spilled, and the recurrence loops
    themselves hold no lane traffic at all -- no v_readlane, no v_writelane: every sample arrives by broadcast."""
    text = "\n".join(demod_asm)
    for name in ("_ZN7airband11tone_kernelILb1EEEvNS_9DemodArgsEii", "_ZN7airband11tone_kernelILb0EEEvNS_9DemodArgsEii"):
        at = text.index(".name:           " + name)
        meta = text[at:at + 1200]
        m = re.search(r"\.sgpr_spill_count:\s+(\d+)", meta)
        v = re.search(r"\.vgpr_spill_count:\s+(\d+)", meta)
        assert m and v, "tone_kernel metadata not found"
        assert int(v.group(1)) == 0, "%s spills %s vector registers" % (name, v.group(1))
        assert int(m.group(1)) <= 16, "%s parks %s scalar registers in vector lanes" % (name, m.group(1))
    body = _function(demod_asm, "tone_kernel")
    assert not any(re.search(r"scratch_(load|store)|buffer_(load|store).*offen", l) for l in body), "tone_kernel uses scratch memory"
    # the recurrence loops: innermost loops (a backward branch to their own label) that read broadcasts and multiply
    blocks, cur, label = [], [], None
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            if cur:
                blocks.append((label, cur))
            cur, label = [], m.group(1)
        cur.append(l)
    if cur:
        blocks.append((label, cur))
    loops = []
    for lab, b in blocks:
        back = [i for i, l in enumerate(b) if lab and re.search(r"s_cbranch_\w+\s+" + re.escape(lab) + r"\b", l)]
        if back:
            b = b[:back[-1] + 1]  # (what follows the back edge up to the next label is the loop's exit)
            if any("ds_read_b128" in l for l in b) and sum(1 for l in b if re.search(r"v_(pk_)?mul_f32", l)) >= 8:
                loops.append(b)
    assert len(loops) >= 2, "steady-state recurrence loops not found"
    for b in loops:
        assert not any(re.search(r"v_(read|write)lane", l) for l in b), "lane traffic inside a recurrence loop"


@pytest.mark.parametrize("kernel", ["demod_kernelILi0ELb0ELi1E", "demod_kernelILi3ELb1ELi1E"], ids=["am", "ctcss_front"])
def test_stable_group_is_one_block_of_four_samples(demod_asm, kernel):
    """Round 3: four samples of a stable wavefront run as one basic block (squelch_fsm.h sq_raw_stable4, demod.hip stable_tail4).  What makes it pay is
    its shape: the four squelch steps sit in ONE basic block, no branch between them (the per-sample path has several per sample).  A squelch step
    has three `not >=` float compares (sample against cap and level, average against cap): twelve of them in one block is the group."""
    body = _function(demod_asm, kernel)
    blocks, cur = [], []
    for l in body:  # basic blocks: cut at labels and after branches
        if re.match(r"^\.LBB", l) and cur:
            blocks.append(cur)
            cur = []
        cur.append(l)
        if re.match(r"^\s*s_(c?branch|endpgm|setpc)", l):
            blocks.append(cur)
            cur = []
    blocks.append(cur)
    most = max(sum(1 for l in b if "v_cmp_nge_f32" in l) for b in blocks)
    assert most >= 12, "no basic block holds four squelch steps (%d `not >=` compares in the fullest one)" % most


@pytest.fixture(scope="module")
def dft_asm(tmp_path_factory):
    if not (os.path.exists(HIPCC) or shutil.which("hipcc")):
        pytest.skip("no hipcc")
    out = str(tmp_path_factory.mktemp("isa") / "dft.s")
    src = os.path.join(ROOT, "rtlsdr-airband_amd", "csrc", "channelizer_dft.hip")
    cmd = [HIPCC, "--offload-arch=gfx950", "-std=c++17"] + _product_flags("channelizer_dft.hip") + ["-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", out, src]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL, timeout=900)
    return open(out).read().split("\n")


def test_channelizer_waits_count_the_stores_it_really_issues(dft_asm):
    """The pipelined loop of the hop-320 channelizer proves that a staging transfer has landed by counting the memory operations issued since -- the
    pieces of later transfers and the OUTPUT STORES (channelizer_dft.hip, `stores` / k_tile: one 16-byte store for |bin|, two for raw I/Q per whole
    tile).  Counting more stores than the machine code issues would let a wait return early, so the shape is pinned: the whole-tile path of the hot loop
    issues at least three 16-byte stores, the loop carries the DMA pieces of one step, 44 MFMAs, 64 sign flips, and its steady-state wait (two steps of
    six pieces + two tiles of three stores = 18) exists as an immediate."""
    body = _function(dft_asm, "channelizer_dft_kernelILi512ELb1ELi320ELb0ELi16ELi1E")
    loop = _hot_loop(body, r"v_mfma_i32_16x16x64_i8")
    assert sum(1 for l in loop if "v_mfma_i32_16x16x64_i8" in l) == 44
    assert sum(1 for l in loop if re.search(r"global_store_dwordx4", l)) >= 3
    assert sum(1 for l in loop if "global_load_lds_dwordx4" in l) >= 6
    assert sum(1 for l in loop if re.match(r"^\s*v_xor_b32", l)) == 64
    assert any(re.search(r"s_waitcnt\s+vmcnt\(18\)", l) for l in loop)
    meta = "\n".join(dft_asm)
    at = meta.index(".name:           _ZN7airband12_GLOBAL__N_122channelizer_dft_kernelILi512ELb1ELi320ELb0ELi16ELi1E")
    assert int(re.search(r"\.vgpr_spill_count:\s+(\d+)", meta[at:at + 1500]).group(1)) == 0
    assert int(re.search(r"\.vgpr_count:\s+(\d+)", meta[at:at + 1500]).group(1)) <= 256  # two waves per SIMD


def test_specialised_demod_kinds_do_not_spill(demod_asm):
    """The four specialised lane-per-channel kinds (AM, NFM, NFM + lowpass, CTCSS front) keep their per-channel state in registers: a change that makes one
    of them spill vector registers (round 3: the stable-group block on the plain NFM kind, 143 of them) is a regression no parity test shows."""
    text = "\n".join(demod_asm)
    for k, ct, w in [(k, ct, w) for k, ct in ((0, 0), (1, 0), (2, 0), (3, 1)) for w in (1, 4)]:  # w: wavefronts per workgroup -- 1, and the regrouped handles' 4 (AIRBAND_HIP_FLAG_REGROUP)
        name = "_ZN7airband12demod_kernelILi%dELb%dELi%dEEEvNS_9DemodArgsEii" % (k, ct, w)
        at = text.index(".name:           " + name)
        meta = text[at:at + 1500]
        assert int(re.search(r"\.vgpr_spill_count:\s+(\d+)", meta).group(1)) == 0, name
        assert int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", meta).group(1)) == 0, name


def test_nfm_kinds_take_the_short_square_root(demod_asm):
    """csrc/exact_math.h: every correctly rounded sqrtf() of the NFM kinds is the short sequence with the compiler's general one (which first scales a
    small argument by 2^32, `v_mul_f32 .., 0x4f800000`) behind a seldom-taken branch -- so exactly half of a kernel's `v_sqrt_f32` sit next to such a
    scaling.  A plain sqrtf() slipping back in (every one scaled) costs 6 vector instructions per square root and shows in no parity test."""
    for k, ct in ((1, 0), (2, 0), (3, 1)):
        body = _function(demod_asm, "demod_kernelILi%dELb%dELi1EEE" % (k, ct))
        n_sqrt = sum(1 for l in body if re.match(r"^\s*v_sqrt_f32", l))
        n_scaled = sum(1 for l in body if re.match(r"^\s*v_mul_f32\w*\s+v\d+, 0x4f800000,", l))
        assert n_sqrt >= 8 and n_scaled * 2 == n_sqrt, (k, n_sqrt, n_scaled)
    # and the two divisions by the lowpass gain are corrected products: FMAs exist in this -ffp-contract=off file only where exact_math.h
    # (or the compiler's own division / square root) asks for one
    body = _function(demod_asm, "demod_kernelILi2ELb0ELi1EEE")
    assert sum(1 for l in body if re.match(r"^\s*v_fma_f32", l)) >= 32


PACKED_F32 = r"^\s*v_pk_(mul|fma|add)_f32"


def test_no_kernel_of_the_library_holds_a_packed_f32_instruction(demod_asm, dft_asm, fft_asm, tmp_path):
    """rtlsdr-airband_amd/_build.py, DEVICE_FLAGS: the library is built without v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32.  While another process runs long
    launches on the same GPU such an instruction now and then leaves lanes 48 - 63 of its result wrong (profiles/r05_event_hunt.md: the CTCSS chain's notch
    filter, the exchange FFT's butterflies; none in any arm once they were gone).  No parity test on a GPU of one's own shows them coming back."""
    for name, asm in (("demod.hip", demod_asm), ("channelizer_dft.hip", dft_asm), ("channelizer_fft.hip", fft_asm)):
        assert not [l for l in asm if re.match(PACKED_F32, l)][:3], name
    for name in ("channelizer_f32.hip", "misc_kernels.hip"):
        out = str(tmp_path / (name + ".s"))
        cmd = [HIPCC, "--offload-arch=gfx950", "-std=c++17"] + _product_flags(name) + ["-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", out,
                                                                                          os.path.join(ROOT, "rtlsdr-airband_amd", "csrc", name)]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL, timeout=900)
        assert not [l for l in open(out).read().split("\n") if re.match(PACKED_F32, l)][:3], name


@pytest.fixture(scope="module")
def fft_asm(tmp_path_factory):
    if not (os.path.exists(HIPCC) or shutil.which("hipcc")):
        pytest.skip("no hipcc")
    out = str(tmp_path_factory.mktemp("isa") / "fft.s")
    src = os.path.join(ROOT, "rtlsdr-airband_amd", "csrc", "channelizer_fft.hip")
    cmd = [HIPCC, "--offload-arch=gfx950", "-std=c++17"] + _product_flags("channelizer_fft.hip") + ["-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", out, src]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL, timeout=900)
    return open(out).read().split("\n")


def test_exchange_fft_kernel_is_shuffle_free_and_keeps_four_waves(fft_asm):
    """channelizer_fft8_kernel (fft_size 256 / 512 / 1024): the 64-point FFT across the lanes goes through the wavefront's LDS buffer -- no ds_bpermute --,
    nothing spills, and the 512-point kernel -- also as the transform of the decimated fft 1024 ... 8192 variants -- keeps four wavefronts per SIMD
    (<= 128 vector registers).  (Until round 5 a complex product was one packed multiply and one packed FMA; see the guard above for why not any more:
    it is now two multiplies and two FMAs -- cmul() in channelizer_fft.hip -- , never the five instructions of the compiler's own form with a negation.)"""
    text = "\n".join(fft_asm)
    for logp, logm, max_vgpr in ((2, 0, 128), (3, 0, 128), (3, 1, 128), (3, 2, 128), (3, 3, 128), (3, 4, 128)):  # fft 256, 512; 1024 ... 8192 as 2 ... 16 decimated 512-point transforms
        body = _function(fft_asm, "channelizer_fft8_kernelILi%dELi%dE" % (logp, logm))
        assert not any("ds_bpermute" in l for l in body), logp
        assert not any(re.match(PACKED_F32, l) for l in body), logp
        n_fma = sum(1 for l in body if re.match(r"^\s*v_fmac?_f32", l))
        assert n_fma >= 28, (logp, n_fma)
        at = text.index(".name:           _ZN7airband12_GLOBAL__N_123channelizer_fft8_kernelILi%dELi%dE" % (logp, logm))
        meta = text[at:at + 1500]
        assert int(re.search(r"\.vgpr_spill_count:\s+(\d+)", meta).group(1)) == 0, logp
        assert int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", meta).group(1)) == 0, logp
        assert int(re.search(r"\.vgpr_count:\s+(\d+)", meta).group(1)) <= max_vgpr, logp
    # the exchange itself: per 512-point hop five rounds of eight 8-byte LDS operations (paired by the compiler into ds_read2 / ds_write2)
    body = _function(fft_asm, "channelizer_fft8_kernelILi3ELi0E")
    assert sum(1 for l in body if re.match(r"^\s*ds_write2?_b64", l)) >= 12


@pytest.fixture(scope="module")
def f32_asm(tmp_path_factory):
    if not (os.path.exists(HIPCC) or shutil.which("hipcc")):
        pytest.skip("no hipcc")
    out = str(tmp_path_factory.mktemp("isa") / "f32.s")
    src = os.path.join(ROOT, "rtlsdr-airband_amd", "csrc", "channelizer_f32.hip")
    cmd = [HIPCC, "--offload-arch=gfx950", "-std=c++17"] + _product_flags("channelizer_f32.hip") + ["-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", out, src]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL, timeout=900)
    return open(out).read()


def test_cf32_layouts_with_immediate_offsets_do_not_spill(f32_asm):
    """Round 6 (profiles/r06_experiments.md H, addendum): the CF32 kernel's fft 2048 variants -- which also run fft 4096 / 8192 -- spilled 48 - 126 registers while every lane kept
    its 32 fragment offsets and 12 parking offsets of the padded image in registers.  The layouts whose offsets are immediates (1: hops of an odd number of samples, 2: padding every
    256 stream bytes, 3: no padding needed) must stay free of spills at every size; layout 0 (per-hop padding) is what is left for the other hops and is allowed its old spills."""
    seen = 0
    for blk in f32_asm.split("  - .agpr_count")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        m = re.search(r"channelizer_f32_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E", name)
        if not m:
            continue
        lay = int(m.group(4))
        if lay == 0:
            continue
        seen += 1
        assert int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1)) == 0, name
        assert int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1)) == 0, name
    assert seen == 24  # 4 fft sizes x 2 tile sizes x 3 layouts
