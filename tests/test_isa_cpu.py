"""ISA gate (CPU suite): the cross-workgroup publications of the shipped gfx950 code objects wait for their stores.

Round-3 review: `scdm_bwd_fused_kernel`'s partner exchange (csrc/scdm_attn.hip `publish_dp`; the gradient of the reference's
`SCDM_Attention.forward`, networks/attention.py:109-121) stored its partial dP rows with agent-scope stores and then bumped the item's
counter after a workgroup barrier -- but on gfx950 the barrier's fence is `s_waitcnt lgkmcnt(0)` only, so the shipped ISA was
`global_store_dword ... sc1 ; s_barrier ; global_atomic_add` with no `vmcnt(0)`: a partner could count the part in and read rows that
had not reached L2.  The fix is an explicit `s_waitcnt vmcnt(0)` in every thread before the barrier.  This test disassembles the
code objects inside the built `libtsg_hip.so` and asserts, for every instantiation of the kernels that publish through
"stores -> barrier -> counter atomic", that a `vmcnt(0)` wait sits between the last agent-scope store and the barrier in front of the
counter's atomic.  No GPU needed (hipcc cross-compiles; llvm-objdump reads the bundle).
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "shufflingvideosfortsg_amd", "libtsg_hip.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"

# kernels whose cross-workgroup protocol is "agent-scope stores, workgroup barrier, one integer atomic on a counter"
PUBLISHERS = ("scdm_bwd_fused_kernel", "boundary_bwd_one_kernel", "gemm_nt_f32s_kernel",
              # round 5: the stream-K weight gradient's tile tickets, the persistent LSTM kernels' start barrier, K4's last-arrival finalisation
              "wgrad_split_kernel", "wgrad_bf16_tr_kernel", "lstm_fwd_persist_kernel", "lstm_fwd_persist_w64_kernel", "lstm_bwd_persist2_kernel",
              "gmd_losses_fwd_kernel")


def _disassemble(tmp_path):
    """-> {mangled kernel name: [instruction text, ...]} for every gfx950 bundle in the shipped library."""
    if not os.path.exists(LIB):
        from shufflingvideosfortsg_amd import build
        build.build()
    work = tmp_path / "isa"
    work.mkdir()
    so = work / "lib.so"
    shutil.copy(LIB, so)
    subprocess.run([OBJDUMP, "--offloading", str(so)], check=True, capture_output=True, cwd=work)   # writes the bundles beside the copy
    kernels = {}
    for f in sorted(os.listdir(work)):
        if "gfx950" not in f:
            continue
        want = subprocess.run(["grep", "-c", "-a", "-E", "|".join(PUBLISHERS), str(work / f)], capture_output=True, text=True)
        if want.stdout.strip() in ("", "0"):
            continue                                     # bundle without a publishing kernel: skip the (slow) disassembly
        txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", str(work / f)], check=True, capture_output=True, text=True).stdout
        cur = None
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                cur = m.group(1)
                kernels[cur] = []
            elif cur is not None and line.startswith("\t"):
                kernels[cur].append(line.split("//")[0].strip())
    return kernels


def _counter_atomics(ins):
    """Indices of integer global atomics (the counters / tickets); float atomics (dw, dbias sums) are not publications."""
    return [i for i, t in enumerate(ins) if re.match(r"global_atomic_(add|inc|or)(_u32|_x2)?\s", t) and "_f32" not in t]


def _check_kernel(name, ins):
    """Every agent-scope (sc1, not the system-scope `sc0 sc1` error sink) store that precedes a counter atomic must be followed by a
    `s_waitcnt` containing vmcnt(0) before the next s_barrier.  Returns the number of publication sites verified."""
    sites = 0
    atomics = _counter_atomics(ins)
    assert atomics, f"{name}: no counter atomic found -- the protocol changed, update this gate"
    stores = [i for i, t in enumerate(ins) if t.startswith("global_store") and re.search(r"\bsc1\b", t) and not re.search(r"\bsc0\b", t)]
    for a in atomics:
        before = [i for i in stores if i < a]
        if not before:
            continue
        last = before[-1]
        barriers = [i for i in range(last, a) if ins[i].startswith("s_barrier")]
        if not barriers:
            continue                                     # this atomic is not a publication of those stores (no barrier between them)
        barrier = barriers[-1]                           # the barrier the counter's atomic sits behind (other work -- the row phase's
        # T-sum epilogue, with barriers of its own -- may lie between the stores and it: round 4 moved the wait behind that epilogue)
        waits = [i for i in range(last + 1, barrier) if ins[i].startswith("s_waitcnt") and "vmcnt(0)" in ins[i]]
        assert waits, (f"{name}: agent-scope store at instruction {last} is followed by s_barrier ({barrier}) and the counter atomic "
                       f"({a}) with no s_waitcnt vmcnt(0) in between:\n  " + "\n  ".join(ins[max(last, barrier - 12):barrier + 1]))
        sites += 1
    return sites


@pytest.fixture(scope="module")
def kernels(tmp_path_factory):
    if not os.path.exists(OBJDUMP):
        pytest.skip("llvm-objdump not available")
    return _disassemble(tmp_path_factory.mktemp("isa"))


def test_k1_backward_publication_waits_for_its_stores(kernels):
    names = [k for k in kernels if "scdm_bwd_fused_kernel" in k]
    assert len(names) >= 16, f"expected every (NP, GATE, MROW, storage) instantiation, found {len(names)}"
    total = 0
    for k in names:
        n = _check_kernel(k, kernels[k])
        assert n >= 1, f"{k}: no publication site recognised (stores -> barrier -> counter atomic)"
        total += n
    assert total >= len(names)


def test_k3_one_launch_backward_publication_waits_for_its_stores(kernels):
    names = [k for k in kernels if "boundary_bwd_one_kernel" in k]
    assert names
    for k in names:
        ins = kernels[k]
        # K3's ticket: plain partial-row stores (write-through is not needed: the explicit wait + the loads' sc1 do the work)
        atomics = _counter_atomics(ins)
        assert atomics, k
        a = atomics[0]
        barrier = max(i for i in range(a) if ins[i].startswith("s_barrier"))
        stores = [i for i in range(barrier) if ins[i].startswith("global_store")]
        assert stores
        last = stores[-1]
        assert any(ins[i].startswith("s_waitcnt") and "vmcnt(0)" in ins[i] for i in range(last + 1, barrier)), k


def test_head_gemm_ticket_waits_for_its_partial_rows(kernels):
    """The fused heads (csrc/gemm_f32s.hip, EpiHead): a head wider than one column tile publishes per-tile partial rows (sc1 stores) and
    takes a ticket -- the same protocol, the same gate."""
    names = [k for k in kernels if "gemm_nt_f32s_kernel" in k and "EpiHead" in k]
    assert len(names) >= 12, names                        # 3 row-tile sizes x (3 activations of K5 + K3)
    for k in names:
        assert _check_kernel(k, kernels[k]) >= 1, k


def test_stream_k_weight_gradient_ticket_waits_for_its_partial_tile(kernels):
    """csrc/wgrad_split.hip (round 5, stream-K): a workgroup that holds a SEGMENT of a tile's contraction publishes its partial accumulators
    (sc1 stores), waits for them (vmcnt(0)), meets at a barrier and takes the tile's ticket; the last arriver reads the other slots with sc1
    loads.  The weight gradients of the path's Linears and of nn.LSTM (networks/RNN.py:31,42) go through it."""
    names = [k for k in kernels if "wgrad_split_kernel" in k or "wgrad_bf16_tr_kernel" in k]
    assert len(names) >= 8, names                         # 4 register-staged variants + 4 LDS-DMA ring shapes
    for k in names:
        ins = kernels[k]
        assert _check_kernel(k, ins) >= 1, k
        a = _counter_atomics(ins)[0]
        after = ins[a:]
        assert any(re.match(r"global_load_dwordx4 .*\bsc1\b", t) for t in after), f"{k}: the last arriver's slot loads are not device-coherent"


def test_persistent_lstm_hand_offs_keep_their_scope_bits(kernels):
    """csrc/lstm.hip (BiLSTM, networks/RNN.py:26-48): the persistent kernels exchange h_t / partial dh between workgroups with the data as
    the flag.  What the protocol needs from the ISA (round-4 review: asserted by comment only):
      * the start barrier counts a workgroup as arrived only after its sentinel / tag marks are acknowledged (sc1 stores -> vmcnt(0) ->
        barrier -> arrival atomic): the publication gate of this file;
      * every poll load is an agent-scope (sc1) 16-byte load, and each poll round ends on s_waitcnt vmcnt(0) before its data is looked at;
      * the exchange stores exist in the write-through (sc1) form (taken whenever a group is not verified to sit on one XCD);
      * a give-up reaches the host: a system-scope (sc0 sc1) store to the error sink, and the launch's error word is polled with an
        sc1 load that is waited for at once."""
    names = [k for k in kernels if "lstm_fwd_persist_kernel" in k or "lstm_fwd_persist_w64_kernel" in k or "lstm_bwd_persist2_kernel" in k]
    assert any("w64" in k for k in names)
    assert len(names) >= 30, len(names)                   # forward: 3 arithmetic modes x 4 hidden sizes x {out-polling, ring} + 4-wave kernels; backward: 12
    for k in names:
        ins = kernels[k]
        # the ARRIVAL atomic is the first global_atomic_add (the XCD-mask `or` sits in front of it; a later add only counts the workgroups
        # on the L2-local path -- a statistic behind the error-raising stores, not a publication)
        arrive = next(i for i, t in enumerate(ins) if re.match(r"global_atomic_add(_u32)?\s", t))
        marks = [i for i in range(arrive) if re.match(r"global_store_dword(x4)? ", ins[i]) and re.search(r"\bsc1\b", ins[i]) and not re.search(r"\bsc0\b", ins[i])]
        assert marks, f"{k}: no sentinel / tag marks in front of the arrival atomic"
        barrier = next((i for i in range(marks[-1], arrive) if ins[i].startswith("s_barrier")), None)
        assert barrier is not None, f"{k}: no workgroup barrier between the marks and the arrival"
        assert any(ins[i].startswith("s_waitcnt") and "vmcnt(0)" in ins[i] for i in range(marks[-1] + 1, barrier)), \
            f"{k}: the marks are not waited for (vmcnt(0)) before the barrier in front of the arrival atomic"
        polls = [i for i, t in enumerate(ins) if re.match(r"global_load_dwordx4 ", t) and re.search(r"\bsc1\b", t)]
        assert len(polls) >= 2, f"{k}: {len(polls)} agent-scope poll loads"
        last = polls[-1]
        assert any(t.startswith("s_waitcnt") and "vmcnt(0)" in t for t in ins[last + 1:last + 4]), f"{k}: poll round does not end on vmcnt(0)"
        wt = [t for t in ins if re.match(r"global_store_dword(x4)? ", t) and re.search(r"\bsc1\b", t) and not re.search(r"\bsc0\b", t)]
        assert len(wt) >= 2, f"{k}: write-through exchange stores missing"
        assert any(t.startswith("global_store_dword ") and re.search(r"\bsc0 sc1\b", t) for t in ins), f"{k}: no system-scope store to the error sink"
        ew = [i for i, t in enumerate(ins) if re.match(r"global_load_dword v\d+, v\[\d+:\d+\], off sc1$", t) and "vmcnt(0)" in ins[i + 1]]
        assert ew, f"{k}: no error-word poll of the form `load sc1 ; s_waitcnt vmcnt(0)` inside the poll loops"


def test_lstm_backward_operand_streams_are_dma_and_nobody_waits_for_them_inside_the_step(kernels):
    """csrc/lstm.hip, round 5 (BiLSTM backward, networks/RNN.py:26-48 through autograd): the streamed operands of the NEXT step (R, c_(t-1), dOut
    tiles) are requested by non-temporal LDS-DMA -- no destination registers -- one step ahead, and retired by the next poll's own vmcnt(0).
    What made three register-prefetch builds SLOWER than loads in front of the poll was a compiler-placed s_waitcnt vmcnt in front of a
    loop-carried copy at the end of the step (profiles/r5/lstm_bwd_operand_dma_v1.txt).  In the shipped code object: every backward kernel
    requests by `global_load_lds_dwordx4 ... nt`, and between the last request of the step loop and the write-through partial-dh stores
    behind it there is no s_waitcnt that mentions vmcnt."""
    names = [k for k in kernels if "lstm_bwd_persist2_kernel" in k]
    assert len(names) >= 12, names
    for k in names:
        ins = kernels[k]
        dma = [i for i, t in enumerate(ins) if t.startswith("global_load_lds_dwordx4")]
        assert len(dma) >= 4, f"{k}: {len(dma)} LDS-DMA requests (prologue + step loop expected)"
        assert all(re.search(r"\bnt\b", ins[i]) for i in dma), f"{k}: an operand DMA without the non-temporal hint"
        first_poll = next(i for i, t in enumerate(ins) if re.match(r"global_load_dwordx4 ", t) and re.search(r"\bsc1\b", t))
        # the partial-dh stores of the step: the first 16-byte write-through stores behind the poll; the loop's requests sit between the two
        # (the prologue's requests are laid out elsewhere: in front of the poll or behind the loop)
        stores = [i for i in range(first_poll, len(ins)) if re.match(r"global_store_dwordx4 ", ins[i]) and re.search(r"\bsc1\b", ins[i])]
        assert stores, f"{k}: no partial-dh stores behind the poll"
        in_loop = [i for i in dma if first_poll < i < stores[0]]
        assert in_loop, f"{k}: no request between the poll and the partial-dh stores"
        between = ins[in_loop[-1] + 1:stores[0]]
        bad = [t for t in between if t.startswith("s_waitcnt") and "vmcnt" in t]
        assert not bad, f"{k}: the step waits for its own operand requests ({bad[0]}) in front of the partial-dh stores"
        assert not any(re.match(r"global_load_(dword|dwordx2|dwordx4|ushort|short_d16) v", t) and " nt" in t for t in between), \
            f"{k}: a streamed REGISTER load inside the step"


def test_k4_last_arrival_finalisation_is_a_release(kernels):
    """csrc/losses.hip (grounding/loss.py:6-51 as train.py:142-165 combines them): every workgroup adds its loss terms with float atomics,
    then `__threadfence()` and a ticket; the last arriver reads the sums.  On gfx950 the fence must be the full release -- buffer_wbl2 +
    s_waitcnt vmcnt(0) -- in front of the ticket's atomic, and an invalidate behind it."""
    names = [k for k in kernels if "gmd_losses_fwd_kernel" in k]
    assert names
    for k in names:
        ins = kernels[k]
        a = _counter_atomics(ins)
        assert a, k
        a = a[0]
        window = ins[max(0, a - 16):a]
        assert any(t.startswith("buffer_wbl2") for t in window) and any(t.startswith("s_waitcnt") and "vmcnt(0)" in t for t in window), \
            f"{k}: no release (buffer_wbl2 + vmcnt(0)) in front of the ticket:\n  " + "\n  ".join(window)
        assert any(t.startswith("buffer_inv") for t in ins[a:a + 24]), f"{k}: no invalidate behind the ticket"


def test_gate_detects_the_round3_sequence():
    """The checker itself: the sequence the round-3 library shipped must fail, the fixed one must pass."""
    bad = ["global_store_dword v[34:35], v38, off sc1", "s_waitcnt lgkmcnt(0)", "s_barrier", "global_atomic_add v34, v35, s[4:5]"]
    good = ["global_store_dword v[34:35], v38, off sc1", "s_waitcnt vmcnt(0)", "s_waitcnt lgkmcnt(0)", "s_barrier",
            "global_atomic_add v34, v35, s[4:5]"]
    with pytest.raises(AssertionError):
        _check_kernel("bad", bad)
    assert _check_kernel("good", good) == 1
