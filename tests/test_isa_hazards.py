"""Compile-time check of the hand-written MFMA streams (CPU: hipcc cross-compiles gfx950 without a GPU).

The XCD-pair recurrences and the bf16 persistent kernels issue their MFMAs from inline asm so that the weight operand can
be an AGPR.  The compiler's hazard recogniser does not look into inline asm, so the kernels write the wait states out by
hand - which is only sound while NOTHING but the asm MFMAs touches an accumulator between the first and the last MFMA of
a stream (a rematerialised zeroing or a register-allocator copy placed there reads or writes a register whose MFMA is
still in flight: this happened once, silently, and produced garbage).  This test disassembles the device code and checks
exactly that, plus the absence of scratch traffic (spills) in those kernels."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

KERNELS = {   # mangled-name fragment -> (MFMA mnemonic, MFMAs per stream, stream is one textual run of code)
    # the XCD-pair kernels, one instantiation per width (N = 128 NKB): a half step's stream is 32 NKB MFMAs
    "lstm_fwd_pair_kernelILi8E": ("v_mfma_f32_16x16x4_f32", 256, True),
    "lstm_bwd_pair_kernelILi8E": ("v_mfma_f32_16x16x4_f32", 256, True),
    "lstm_fwd_pair_kernelILi7E": ("v_mfma_f32_16x16x4_f32", 224, True),
    "lstm_bwd_pair_kernelILi7E": ("v_mfma_f32_16x16x4_f32", 224, True),
    "lstm_fwd_pair_kernelILi6E": ("v_mfma_f32_16x16x4_f32", 192, True),
    "lstm_bwd_pair_kernelILi6E": ("v_mfma_f32_16x16x4_f32", 192, True),
    "lstm_fwd_pair_kernelILi5E": ("v_mfma_f32_16x16x4_f32", 160, True),
    "lstm_bwd_pair_kernelILi5E": ("v_mfma_f32_16x16x4_f32", 160, True),
    # the bf16 kernels' step-0 path (no product: accumulators zeroed) is laid out between the MFMA blocks of the other
    # path, so the textual accumulator check does not apply to them; spills, copies and operand classes do
    "lstm_fwd_persist_bf16_kernelILi8ELi2ELb0": ("v_mfma_f32_16x16x32_bf16", 64, False),
    "lstm_bwd_persist_bf16_kernelILi4ELi2ELb0": ("v_mfma_f32_16x16x32_bf16", 64, False),
}


def _regs(text):
    used = set()
    for m in re.finditer(r"\bv(\d+)\b", text):
        used.add(int(m.group(1)))
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        used.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return used


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "lstm.s"
    src = os.path.join(ROOT, "lstm_ctc_amd", "csrc", "lstm.hip")
    subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", str(out), src],
                   check=True, capture_output=True, timeout=900)
    return open(out).read().split("\n")


def test_no_spills_in_any_persistent_kernel(device_asm):
    """Every instantiation of the persistent recurrences (single-XCD fp32 / bf16, XCD pair) and their verify kernel:
    no scratch access anywhere - a spill reload inside a time loop is a vector-memory instruction in a latency chain."""
    cur, bad = None, {}
    for l in device_asm:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur = m.group(1)
        elif cur and ("persist" in cur or "pair_kernel" in cur) and "scratch_" in l:
            bad[cur] = bad.get(cur, 0) + 1
    assert not bad, bad


@pytest.mark.parametrize("kernel", sorted(KERNELS))
def test_nothing_touches_an_accumulator_inside_an_mfma_stream(device_asm, kernel):
    mnemonic, per_stream, contiguous = KERNELS[kernel]
    start = next(i for i, l in enumerate(device_asm) if re.match(r"^_Z\w*%s\w*:" % kernel, l))
    end = next(i for i in range(start, len(device_asm)) if device_asm[i].strip() == "s_endpgm")
    body = device_asm[start:end]
    assert not any("scratch_" in l for l in body), "spill code in %s" % kernel
    assert not any(l.strip().startswith("v_accvgpr") for l in body), "AGPR <-> VGPR copies in %s" % kernel
    mfma = [i for i, l in enumerate(body) if l.strip().startswith(mnemonic)]
    assert mfma and len(mfma) % per_stream == 0, (kernel, len(mfma))
    for s0 in range(0, len(mfma), per_stream):
        stream = mfma[s0:s0 + per_stream]
        acc = set()
        for i in stream:
            m = re.match(r"\s*%s v\[(\d+):(\d+)\], (\S+), (a\[?\d+)" % mnemonic, body[i])
            assert m, body[i]                                  # the weight operand is an AGPR, the destination a VGPR tuple
            acc.update(range(int(m.group(1)), int(m.group(2)) + 1))
        if not contiguous:
            continue
        for i in range(stream[0], stream[-1]):
            t = body[i].strip()
            if not t or t[0] in ";." or t.startswith(mnemonic) or t.startswith("s_"):
                continue
            assert not (_regs(t) & acc), "%s: `%s` touches an accumulator inside an MFMA stream" % (kernel, t)


X3_KERNELS = {   # split-operand XCD-pair kernels: MFMAs per half-step stream, of which with the Rl operand in VGPRs (from LDS)
    "lstm_fwd_pair_x3_kernelILi8E": (192, 32),
    "lstm_fwd_pair_x3_kernelILi6E": (144, 24),
    "lstm_bwd_pair_x3_kernelILi8E": (192, 32),
    "lstm_bwd_pair_x3_kernelILi6E": (144, 24),
}


@pytest.mark.parametrize("kernel", sorted(X3_KERNELS))
def test_split_operand_pair_kernels(device_asm, kernel):
    """lstm_pair_x3.inc: v_mfma_f32_16x16x32_bf16 from inline asm - Rh / Rm operands are AGPR tuples, the Rl operand a VGPR
    tuple (VGPR-resident or read from LDS); nothing but the MFMAs touches an accumulator between the first and the last MFMA
    of a half step's stream (the retry loops of the freshness checks sit inside it textually and must keep clear too); no
    scratch, no AGPR <-> VGPR copies; and - what the stream's pace depends on - no `s_waitcnt vmcnt(0)` on its main path: the
    only ones allowed are those of the asm retry loads (each directly behind a buffer_load of a retry loop)."""
    per_stream, n_vgpr_b = X3_KERNELS[kernel]
    mnemonic = "v_mfma_f32_16x16x32_bf16"
    start = next(i for i, l in enumerate(device_asm) if re.match(r"^_Z\w*%s\w*:" % kernel, l))
    end = next(i for i in range(start, len(device_asm)) if device_asm[i].strip() == "s_endpgm")
    body = device_asm[start:end]
    assert not any("scratch_" in l for l in body), "spill code in %s" % kernel
    assert not any(l.strip().startswith("v_accvgpr") for l in body), "AGPR <-> VGPR copies in %s" % kernel
    mfma = [i for i, l in enumerate(body) if l.strip().startswith(mnemonic)]
    # (textually there are up to four streams - (group, first half step or not) - but the compiler merges identical tails of
    # some: the walk below is local, over textually consecutive MFMAs, and does not rely on whole streams)
    assert len(mfma) > 2 * per_stream, (kernel, len(mfma))
    dsts, from_vgpr = [], 0
    for i in mfma:
        m = re.match(r"\s*%s v\[(\d+):(\d+)\], v\[\d+:\d+\], ([av])\[\d+:\d+\], v\[(\d+):(\d+)\]" % mnemonic, body[i])
        assert m and m.group(1) == m.group(4) and m.group(2) == m.group(5), body[i]          # D == C: an accumulate chain
        dsts.append(set(range(int(m.group(1)), int(m.group(2)) + 1)))
        from_vgpr += m.group(3) == "v"
    # one product in six takes Rl from VGPRs (exactly, where no tails were merged)
    assert abs(from_vgpr * per_stream - n_vgpr_b * len(mfma)) <= 0.1 * n_vgpr_b * len(mfma), (kernel, from_vgpr, len(mfma))
    assert len(mfma) != 4 * per_stream or from_vgpr == 4 * n_vgpr_b
    checked = 0
    for k in range(len(mfma) - 1):
        gap = [body[i].split(";")[0].strip() for i in range(mfma[k] + 1, mfma[k + 1])]
        if any(t.startswith("s_nop 15") or t == "s_nop 7" or t.startswith("s_barrier") or t.startswith("s_endpgm") for t in gap):
            continue              # a stream's end (s_nop 15 .., tile stores, barrier) or start (accumulators zeroed, s_nop 7)
        acc = set().union(*dsts[max(0, k - 7):k + 1])        # every accumulator comes round within eight MFMAs
        for j, t in enumerate(gap):
            if not t or t[0] == "." or t.endswith(":") or t.startswith("s_"):
                continue
            assert not (_regs(t) & acc), "%s: `%s` touches an accumulator inside an MFMA stream" % (kernel, t)
            checked += 1
        for j, t in enumerate(gap):                 # (BPTT: operand refills are in flight all along; the forward kernel's
            if "bwd" in kernel and t.startswith("s_waitcnt") and "vmcnt(0)" in t:        # only in-stream loads are the receipt's)
                prev = next((x for x in reversed(gap[:j]) if x), "")
                assert prev.startswith("buffer_load") or prev.startswith("global_load"), (kernel, gap[max(0, j - 3):j + 1])
    assert checked > per_stream


X3_PERSIST = {"lstm_fwd_persist_x3_kernelILi%dE" % nb: ((nb + 3) // 4) ** 2 * 6 for nb in range(2, 17, 2)}
X3_PERSIST.update({"lstm_bwd_persist_x3_kernelILi%dE" % nbw: nbw * 6 for nbw in range(2, 17, 2)})


@pytest.mark.parametrize("kernel", sorted(X3_PERSIST))
def test_split_operand_single_xcd_kernels(device_asm, kernel):
    """The single-XCD split-operand recurrences (N = 64 .. 512): one textual MFMA stream per time step - PB x NT x 6 products
    forward, NBW x 6 in the BPTT -, every MFMA an accumulate chain (D == C) with its R term in an AGPR tuple (five of six) or a
    VGPR tuple (Rl), no scratch, no AGPR <-> VGPR copies inside the time loop, and between the first and the last MFMA of the stream nothing but
    MFMAs touches an accumulator (the BPTT's chunk loads and their retry loops sit inside it)."""
    mnemonic = "v_mfma_f32_16x16x32_bf16"
    start = next(i for i, l in enumerate(device_asm) if re.match(r"^_Z\w*%s\w*:" % kernel, l))
    end = next(i for i in range(start, len(device_asm)) if device_asm[i].strip() == "s_endpgm")
    body = device_asm[start:end]
    assert not any("scratch_" in l for l in body), "spill code in %s" % kernel
    mfma = [i for i, l in enumerate(body) if l.strip().startswith(mnemonic)]
    assert len(mfma) == X3_PERSIST[kernel], (kernel, len(mfma))
    # the time loop: from the earliest target of a backward branch that jumps over the first MFMA (the prologue that builds the weight
    # fragments may shuffle values through AGPRs at the widest instantiations; the loop must not)
    head = None
    for i in range(mfma[0], len(body)):          # (the latch may sit textually inside the stream: block placement)
        m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", body[i])
        if m:
            j = next((j for j, l in enumerate(body) if l.startswith(m.group(1) + ":")), None)
            if j is not None and j < mfma[0]:
                head = j if head is None else min(head, j)
    assert head is not None, "no time loop found in %s" % kernel
    assert not any(l.strip().startswith("v_accvgpr") for l in body[head:]), "AGPR <-> VGPR copies in the time loop of %s" % kernel
    acc, from_vgpr = set(), 0
    for i in mfma:
        m = re.match(r"\s*%s v\[(\d+):(\d+)\], v\[\d+:\d+\], ([av])\[\d+:\d+\], v\[(\d+):(\d+)\]" % mnemonic, body[i])
        assert m and m.group(1) == m.group(4) and m.group(2) == m.group(5), body[i]
        acc |= set(range(int(m.group(1)), int(m.group(2)) + 1))
        from_vgpr += m.group(3) == "v"
    assert from_vgpr * 6 == len(mfma), (kernel, from_vgpr)
    # inside every basic block that holds MFMAs (labels and branches end a block: ragged N / 32 puts `slot < blocks of this wave`
    # tests between the slots, and block placement may park other paths textually between them): nothing but MFMAs touches
    # an accumulator between the block's first and last MFMA
    blocks, cur = [], []
    for i in range(mfma[0], mfma[-1] + 1):
        t = body[i].split(";")[0].strip()
        if t.endswith(":") or t.startswith("s_cbranch") or t.startswith("s_branch"):
            blocks.append(cur)
            cur = []
        elif t and t[0] != ".":
            cur.append(t)
    blocks.append(cur)
    seen = 0
    for blk in blocks:
        at = [j for j, t in enumerate(blk) if t.startswith(mnemonic)]
        if not at:
            continue
        seen += len(at)
        for t in blk[at[0]:at[-1] + 1]:
            if t.startswith("s_") or t.startswith(mnemonic):
                continue
            assert not (_regs(t) & acc), "%s: `%s` touches an accumulator inside the MFMA stream" % (kernel, t)
    assert seen == len(mfma)


@pytest.fixture(scope="module")
def ctc_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa_ctc") / "ctc.s"
    src = os.path.join(ROOT, "lstm_ctc_amd", "csrc", "ctc.hip")
    subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", str(out), src],
                   check=True, capture_output=True, timeout=900)
    return open(out).read().split("\n")


def test_no_spills_in_any_ctc_kernel(ctc_asm):
    """Every CTC kernel the library can launch - all instantiations of the meet-in-the-middle scan and the three-kernel
    path: no scratch instruction.  (Round 2 shipped ctc_mm_kernel<8,4,*,2> with ~830 of them; lattices of more than 1024
    positions now take the three-kernel path, whose 8-positions-per-lane scan fits the register file.)  Also pins the
    geometry the B = 512 roofline figure is measured on to two waves per SIMD."""
    cur, spills, vgprs = None, {}, {}
    for l in ctc_asm:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur = m.group(1)
        elif cur and "scratch_" in l and not l.strip().startswith(";"):
            spills[cur] = spills.get(cur, 0) + 1
        elif cur:
            m = re.match(r"\s*;\s*NumVgprs:\s*(\d+)", l)
            if m:
                vgprs[cur] = int(m.group(1))
    kernels = [k for k in vgprs if "ctc_" in k]
    assert len(kernels) >= 20 and any("ctc_mm_kernelILi4ELi1ELi3ELi2" in k for k in kernels)
    assert not spills, spills
    assert not any("ctc_mm_kernelILi8" in k for k in kernels)          # the spilling instantiation is gone, not hidden
    wide = [k for k in kernels if "ctc_mm_kernelILi4ELi1E" in k]
    assert wide and all(vgprs[k] <= 256 for k in wide)


@pytest.fixture(scope="module")
def x3_asm(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa_x3") / "gemm_x3.s"
    src = os.path.join(ROOT, "lstm_ctc_amd", "csrc", "gemm_x3.hip")
    subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", str(out), src],
                   check=True, capture_output=True, timeout=600)
    return open(out).read().split("\n")


def test_bf16x3_kernels_keep_their_pipeline(x3_asm):
    """The bf16x3 product kernels as the compiler emits them: no scratch, at most 256 VGPRs (two waves per SIMD: one
    workgroup of 8 waves per CU), 48 MFMAs per k tile, and - what the three-stage NT kernel lives on - NO `s_waitcnt
    vmcnt(0)` inside its k loop except on the branch that ends the walk: the fill of tile t + 2 must stay in flight across
    the barrier of tile t (a C++-level LDS read or a __syncthreads fence in that loop would make the compiler drain it)."""
    body, cur = {}, None
    for l in x3_asm:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur = m.group(1)
            body[cur] = []
        elif cur:
            body[cur].append(l)
    nt = next(k for k in body if "gemm_x3_kernel" in k)
    tn = next(k for k in body if "gemm_x3_tn_kernel" in k)
    for k in (nt, tn):
        text = [l.split(";")[0] for l in body[k]]
        assert not any("scratch_" in l for l in text), k
        nv = [int(m.group(1)) for l in body[k] for m in [re.match(r"\s*;\s*NumVgprs:\s*(\d+)", l)] if m]
        assert nv and nv[0] <= 256, (k, nv)
        assert sum("v_mfma_f32_32x32x16_bf16" in l for l in text) == 48, k       # one k tile's worth, once (a rolled loop)
    # the NT loop: from the first fragment read to the loop's barrier
    text = [l.split(";")[0].strip() for l in body[nt]]
    first = next(i for i, l in enumerate(text) if l.startswith("ds_read_b128"))
    last = max(i for i, l in enumerate(text) if l.startswith("v_mfma")) + 16          # ... to the waits behind the last MFMA
    drains = [l for l in text[first:last] if l.startswith("s_waitcnt") and "vmcnt(0)" in l]
    counted = [l for l in text[first:last] if l.startswith("s_waitcnt") and "vmcnt(6)" in l]
    assert len(drains) <= 1 and len(counted) == 1, (drains, counted)
    assert sum(l.startswith("ds_read_b128") for l in text[first:last]) == 18
    assert not any(l.startswith("s_waitcnt") and "vmcnt" in l for l in text[first:last - 16])     # none among the reads / MFMAs
    # the TN kernel's fragments come from transposing reads only
    ttext = [l.split(";")[0].strip() for l in body[tn]]
    assert sum(l.startswith("ds_read_b64_tr_b16") for l in ttext) == 36 and not any(l.startswith("ds_read_b128") for l in ttext)
