"""The counted `s_waitcnt vmcnt(N)` immediates of the hand-scheduled full-row kernels (csrc/gemm_frd.hip, csrc/gemm_fr64.hip)
are compile-time constants derived by hand from the kernels' issue order.  A count that is too LARGE lets an MFMA read a
fragment that has not landed (wrong results, timing-dependent); too small only costs time.  This CPU test re-derives every
one of them from an independent simulation of the per-wave vector-memory queue (operations retire in issue order, so "what
may stay in flight when X must have landed" = the number of operations issued after X) and compares with the source:

* gemm_frd.hip: `frd_vm(j, nb, last)` is extracted from the .hip file, compiled with g++ and evaluated for every (slab position,
  fragment, last-slab) — against a simulation of whole K loops of 1 .. 6 slabs;
* gemm_fr64.hip: the VM0 / VM1 / VMT4 / VMT3 table of FH<6> (three-slot ring) and FH<8> (four-slot ring) parsed from the
  source — against a simulation of its stage schedule for K loops of 2 .. 8 slabs.
"""
import os
import re
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "ditto_tts_amd", "csrc")


# ---------------------------------------------------------------- gemm_frd.hip
def _frd_vm_from_source():
    s = open(os.path.join(CSRC, "gemm_frd.hip")).read()
    body = s[s.index("constexpr int frd_vm(int j, int nb, bool last) {"):]
    body = body[:body.index("\n}\n") + 3]
    prog = "#include <cstdio>\n" + body + """
int main() {
    for (int last = 0; last < 2; ++last) for (int j = 0; j < 4; ++j) for (int nb = 0; nb < 6; ++nb)
        std::printf("%d %d %d %d\\n", last, j, nb, frd_vm(j, nb, last != 0));
    return 0;
}
"""
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "v.cpp"), os.path.join(d, "v")
        open(src, "w").write(prog)
        subprocess.run(["g++", "-std=c++17", "-O0", src, "-o", exe], check=True)
        out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    return {(int(a), int(b), int(c)): int(v) for a, b, c, v in (ln.split() for ln in out.strip().splitlines())}


def _simulate_frd(nslab):
    """Issue order of one wave of gemm_frd_kernel (header + stage lambda): prologue W(0,*), W(1,*) with nothing behind them;
    stage s at slab position j issues, per fragment nb: [wait W(s, nb)] 4 MFMAs, W(s+2, nb) while stage s+2 exists, and at
    j < 2 while a next slab exists an A piece behind nb = 4 and nb = 5.  Returns {(last, j, nb): set of observed counts}."""
    nkt = 4 * nslab
    queue = []                                     # issue-ordered list of operation names
    for s in (0, 1):
        for nb in range(6):
            queue.append(("W", s, nb))
    seen = {}
    for s in range(nkt):
        j, slab = s & 3, s >> 2
        last = slab == nslab - 1
        for nb in range(6):
            younger = len(queue) - 1 - queue.index(("W", s, nb))
            seen.setdefault((int(last), j, nb), set()).add(younger)
            if s + 2 < nkt:
                queue.append(("W", s + 2, nb))
            if not last and j < 2 and nb >= 4:
                queue.append(("A", slab + 1, 2 * j + (nb - 4)))
    return seen


def test_frd_wait_counts_match_a_queue_simulation():
    vm = _frd_vm_from_source()
    for nslab in range(1, 7):
        seen = _simulate_frd(nslab)
        for key, counts in seen.items():
            # the first slab's early stages see FEWER younger operations than the steady state only where the fragment was
            # loaded by the prologue; the constant must never exceed what was really issued after the fragment
            assert vm[key] <= min(counts) or (key[0] == 0 and min(counts) < max(counts) and vm[key] == max(counts)), (nslab, key, counts, vm[key])
            assert vm[key] == max(counts) or vm[key] == min(counts), (nslab, key, counts, vm[key])
    # steady state (a middle slab of a long loop) must be exact: anything smaller would stall the wave for nothing
    seen = _simulate_frd(6)
    for j in range(4):
        for nb in range(6):
            assert vm[(0, j, nb)] == max(seen[(0, j, nb)])
            assert vm[(1, j, nb)] == min(seen[(1, j, nb)]) == max(seen[(1, j, nb)])


def test_frd_first_slab_is_never_too_loose():
    """In the first slab some fragments have fewer operations behind them than in the steady state (no A pieces were issued in
    the two prologue 'stages').  A too-large immediate there would be a real bug: check every stage of every loop length."""
    vm = _frd_vm_from_source()
    for nslab in range(1, 7):
        nkt = 4 * nslab
        queue = [("W", s, nb) for s in (0, 1) for nb in range(6)]
        for s in range(nkt):
            j, slab = s & 3, s >> 2
            last = slab == nslab - 1
            for nb in range(6):
                younger = len(queue) - 1 - queue.index(("W", s, nb))
                assert vm[(int(last), j, nb)] <= younger, (nslab, s, nb, vm[(int(last), j, nb)], younger)
                if s + 2 < nkt:
                    queue.append(("W", s + 2, nb))
                if not last and j < 2 and nb >= 4:
                    queue.append(("A", slab + 1, 2 * j + (nb - 4)))
    # ... and the wait that certifies the next slab's A pieces at (j = 3, nb = 4): vmcnt(10) in the source
    src = open(os.path.join(CSRC, "gemm_frd.hip")).read()
    assert 'asm volatile("s_waitcnt vmcnt(10)" ::: "memory");' in src
    for nslab in range(2, 6):
        nkt = 4 * nslab
        queue = [("W", s, nb) for s in (0, 1) for nb in range(6)]
        for s in range(nkt):
            j, slab = s & 3, s >> 2
            last = slab == nslab - 1
            for nb in range(6):
                if j == 3 and nb == 4 and not last:
                    pieces = [i for i, op in enumerate(queue) if op[0] == "A" and op[1] == slab + 1]
                    assert len(pieces) == 4 and len(queue) - 1 - max(pieces) >= 10, (nslab, s)
                if s + 2 < nkt:
                    queue.append(("W", s + 2, nb))
                if not last and j < 2 and nb >= 4:
                    queue.append(("A", slab + 1, 2 * j + (nb - 4)))


# ---------------------------------------------------------------- gemm_fr64.hip
def _fr64_table():
    s = open(os.path.join(CSRC, "gemm_fr64.hip")).read()
    m0 = re.search(r"VM0 = NBW == 6 \? (\d+) : (\d+), VM1 = NBW == 6 \? (\d+) : (\d+);", s)
    mt = re.search(r"VMT4 = NBW == 6 \? (\d+) : (\d+), VMT3 = NBW == 6 \? (\d+) : (\d+);", s)
    t4 = re.search(r"T4_ISSUES_W = NBW == 6 \? (\d) : (\d);", s)
    a, b, c, d = (int(x) for x in m0.groups())
    e, f, g, h = (int(x) for x in mt.groups())
    return {6: dict(VM0=a, VM1=c, VMT4=e, VMT3=g, T4W=int(t4.group(1))), 8: dict(VM0=b, VM1=d, VMT4=f, VMT3=h, T4W=int(t4.group(2)))}


def _simulate_fr64(nbw, hns, nslab):
    """One wave of gemm_fr64_kernel<NBW>: prologue A slabs 0, 1, W stages 0 .. hns-2, then (behind the accumulator init and a
    full drain) W stage hns-1.  Stage s (j = s & 1): W piece nb of stage s + hns behind MFMA pair nb while that stage exists;
    at nb = NBW - 3: the counted wait for W(s+1, all) and, at j = 1, for this wave's piece of slab (s+1)/2, then (j = 1, while
    slab (s+1)/2 + 1 exists) the A piece of that slab.  Returns the number of operations younger than the LAST needed one at
    every wait, keyed by stage."""
    nkt = 2 * nslab
    queue = [("W", hns - 1, nb) for nb in range(nbw)]      # everything older was drained by vmcnt(0) before the loop
    landed = {("W", s) for s in range(hns - 1)} | {("A", 0), ("A", 1)}
    out = {}
    for s in range(nkt):
        j = s & 1
        for nb in range(nbw):
            if nb == nbw - 3 and s + 1 < nkt:
                need = []
                if ("W", s + 1) not in landed:
                    need.append(max(i for i, op in enumerate(queue) if op[0] == "W" and op[1] == s + 1))
                if j == 1 and ("A", (s + 1) // 2) not in landed:
                    need.append(max(i for i, op in enumerate(queue) if op[0] == "A" and op[1] == (s + 1) // 2))
                out[s] = (len(queue) - 1 - max(need)) if need else None
                if j == 1 and (s + 1) // 2 + 1 < nslab:
                    queue.append(("A", (s + 1) // 2 + 1, 0))
            if s + hns < nkt:
                queue.append(("W", s + hns, nb))
    return out


def test_fr64_wait_table_matches_a_queue_simulation():
    tab = _fr64_table()
    for nbw, hns in ((6, 3), (8, 4)):
        t = tab[nbw]
        assert t["T4W"] == (1 if hns == 3 else 0)
        for nslab in range(2, 9):
            nkt = 2 * nslab
            sim = _simulate_fr64(nbw, hns, nslab)
            for s, younger in sim.items():
                if s < nkt - 4:
                    vm = t["VM0"] if s % 2 == 0 else t["VM1"]
                elif s == nkt - 4:
                    vm = t["VMT4"]
                elif s == nkt - 3:
                    vm = t["VMT3"]
                else:
                    vm = 0
                if younger is None:
                    continue          # everything it needs landed before the loop: any count is safe
                assert vm <= younger, (nbw, nslab, s, vm, younger)
                # exact in the steady state (stages that are neither among the first hns nor in the tail)
                if hns <= s < nkt - 4:
                    assert vm == younger, (nbw, nslab, s, vm, younger)


# ---------------------------------------------------------------- gemm_frd.hip: the asynchronous register ring in the ISA
def test_frd_register_ring_is_never_copied_while_loads_are_in_flight():
    """gemm_frd's W fragments are loaded by asm `global_load_dwordx4` statements into a register ring that hipcc believes is
    written synchronously; the kernel's own counted waits make that true before each use.  What would break it silently is a
    compiler-inserted COPY (v_mov / v_accvgpr_write) or a spill of a ring register between a load's issue and its wait.  This
    test compiles the file to ISA (device pass only, a few seconds) and checks, for all seven instantiations, that from the
    first to the last MFMA no instruction reads a ring register except MFMAs and no scratch access exists at all."""
    src = os.path.join(CSRC, "gemm_frd.hip")
    inc = os.path.join(HERE, "..", "include")
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "frd.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", inc, "-I", CSRC,
                        "--cuda-device-only", "-S", src, "-o", out], check=True, capture_output=True)
        text = open(out).read()
    parts = re.split(r"\n(_ZN5ditto12_GLOBAL__N_115gemm_frd_kernel\w+): ; @", text)
    assert len(parts) == 15, ("seven instantiations expected (<LN, RES> on the fp32 stream, <LN, true> on the bf16 stream, and the "
                              "plain product with a bf16 result of the training backward)")
    for i in range(1, len(parts), 2):
        body = parts[i + 1].split("s_endpgm")[0].split("\n")
        loads = [(n, re.match(r"\s*global_load_dwordx4 v\[(\d+):(\d+)\], v\d+, s\[", ln)) for n, ln in enumerate(body)]
        loads = [(n, m) for n, m in loads if m]
        assert len(loads) >= 24, parts[i]
        ring = set()
        for _, m in loads:
            ring.update(range(int(m.group(1)), int(m.group(2)) + 1))
        assert len(ring) == 48, (parts[i], len(ring))                     # two stages x six fragments x four registers
        assert not any("scratch_" in ln for ln in body), parts[i]
        # walk the instruction stream: a register is IN FLIGHT from the W load that targets it to the first MFMA that reads it
        # (the kernel's counted wait sits right in front of that MFMA); nothing else may touch it in between
        inflight, nload, nuse = set(), 0, 0
        for n, raw in enumerate(body):
            ln = raw.split(";")[0].strip()
            if not ln or ln.startswith(".") or ln.endswith(":"):
                continue
            m = re.match(r"(\S+)\s*(.*)", ln)
            op, args = m.group(1), m.group(2)
            regs = set()
            for x, y in re.findall(r"v\[(\d+):(\d+)\]", args):
                regs.update(range(int(x), int(y) + 1))
            regs.update(int(x) for x in re.findall(r"\bv(\d+)\b", args))
            wl = re.match(r"global_load_dwordx4 v\[(\d+):(\d+)\], v(\d+), s\[", ln)
            if wl:
                dst = set(range(int(wl.group(1)), int(wl.group(2)) + 1))
                assert not (dst & inflight), (parts[i][-28:], n, ln, "loaded into a register whose previous load was never consumed")
                assert int(wl.group(3)) not in inflight, (parts[i][-28:], n, ln, "address register is an in-flight destination")
                inflight |= dst
                nload += 1
            elif op.startswith("v_mfma"):
                hit = regs & inflight
                if hit:
                    nuse += 1
                    inflight -= hit
            else:
                assert not (regs & inflight), (parts[i][-28:], n, ln, "touches a W fragment register while its load is in flight")
        assert nload >= 24 and nuse >= 24 and not inflight, (parts[i][-28:], nload, nuse, sorted(inflight))
