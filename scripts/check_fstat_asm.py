"""Static guard for fused_fstat_kernel's hand-counted LDS reads (recompute.hip): hipcc does not know that the destination of
an inline-asm ds_read is still in flight until the asm s_waitcnt that retires it, so it may copy / read that register early
(seen once: a phi copy at a branch between the ring's first reads and the loop).  Compiles the file to ISA and fails if any
compiler-generated vector instruction reads such a register.  Usage: python scripts/check_fstat_asm.py  (exit code 1 = violation)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def vregs(tok):
    out = set()
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def scan_mfma_hazards(asm_text, min_mfmas=4):
    """hipcc does not know the asm statements are MFMAs, so it inserts no wait states between an MFMA and a VALU instruction that
    reads its accumulator, and the hardware does not interlock.  Every v_* instruction (compiler-generated or an
    in-asm filler) that reads a VGPR accumulator must come >= min_mfmas MFMAs (>= 128 cycles) or a pair of `s_nop 7` pads after
    the last MFMA that wrote it.  Straight-line approximation: ages are tracked in file order within a kernel."""
    bad = []
    for m in re.finditer(r'^(_Z\d+fused_fstat_kernel\w*):', asm_text, re.M):
        body = asm_text[m.end():asm_text.index('s_endpgm', m.end())].split('\n')
        age = {}  # vgpr -> MFMAs issued since an MFMA last wrote it
        for ln, l in enumerate(body):
            t = l.strip()
            if not t or t[0] in ';.' or t.endswith(':') or 'ASM' in t:
                continue
            if t.startswith('v_mfma'):
                ops = t.split(None, 1)[1].split(',')
                dst = vregs(ops[0])
                for r in age:
                    age[r] += 1
                for r in dst:
                    age[r] = 0
                continue
            if t.startswith('s_nop 7'):
                for r in age:
                    age[r] += 2  # 8 wait states; two of them in a row clear the hazard
                continue
            if t.startswith('v_') and ',' in t:  # sources only: a pure overwrite of a stale accumulator register is harmless
                regs = vregs(','.join(t.split(None, 1)[1].split(',')[1:]))
                hot = [r for r in regs if age.get(r, 99) < min_mfmas]
                if hot:
                    bad.append((m.group(1), ln, t, hot[:4]))
            if t.startswith(('ds_read', 'global_load', 'v_')):  # an overwrite by something else ends the tracking of that register
                for r in vregs(t.split(None, 1)[1].split(',')[0]):
                    age.pop(r, None)
    return bad


def aregs(tok):
    out = set()
    for m in re.finditer(r'\ba\[(\d+):(\d+)\]|\ba(\d+)\b', tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def scan_async_loads(asm_text):
    """The feature refill of the last score visit is an asm global_load into AGPRs; hipcc does not know the destination is in flight
    until the next `s_waitcnt vmcnt(0)`.  Any instruction (MFMA or compiler-generated) that reads such an AGPR before that wait, and
    any v_accvgpr shuffle in the kernel at all (the unit loop is written so that none is needed), is a violation."""
    bad = []
    for m in re.finditer(r'^(_Z\d+fused_fstat_kernel\w*):', asm_text, re.M):
        body = asm_text[m.end():asm_text.index('s_endpgm', m.end())].split('\n')
        flying, inasm = set(), False
        for ln, l in enumerate(body):
            t = l.strip()
            if 'ASMSTART' in t:
                inasm = True
                continue
            if 'ASMEND' in t:
                inasm = False
                continue
            if not t or t[0] in ';.' or t.endswith(':'):
                continue
            if t.startswith('v_accvgpr'):
                bad.append((m.group(1), ln, t))
                continue
            if t.startswith('s_waitcnt') and 'vmcnt(0)' in t:
                flying = set()
                continue
            ops = t.split(None, 1)[1] if ' ' in t else ''
            if inasm and t.startswith('global_load_dwordx4 a'):
                flying |= aregs(ops.split(',')[0])
                continue
            srcs = aregs(','.join(ops.split(',')[1:]))
            if srcs & flying:
                bad.append((m.group(1), ln, t))
    return bad


def scan(asm_text):
    bad, kernels = [], 0
    for m in re.finditer(r'^(_Z\d+fused_fstat_kernel\w*):', asm_text, re.M):
        kernels += 1
        body = asm_text[m.end():asm_text.index('s_endpgm', m.end())].split('\n')
        pending, inasm = [], False
        for ln, l in enumerate(body):
            t = l.strip()
            if 'ASMSTART' in t:
                inasm = True
                continue
            if 'ASMEND' in t:
                inasm = False
                continue
            if not t or t[0] in ';.' or t.endswith(':'):
                continue
            if inasm:
                if t.startswith('ds_read_b128'):
                    pending.append(vregs(t.split(',')[0]))
                elif t.startswith('s_waitcnt') and 'lgkmcnt' in t:
                    n = int(re.search(r'lgkmcnt\((\d+)\)', t).group(1))
                    pending = pending[-n:] if n > 0 else []
            elif t.startswith('v_') and ',' in t:
                srcs = vregs(','.join(t.split(None, 1)[1].split(',')[1:]))
                if any(srcs & p for p in pending):
                    bad.append((m.group(1), ln, t))
    return kernels, bad


def main():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "recompute.s")
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
               "-fhip-fp32-correctly-rounded-divide-sqrt", "-S", "--cuda-device-only", os.path.join(ROOT, "leann-rs_amd/csrc/recompute.hip"), "-o", out]
        subprocess.run(cmd, check=True, capture_output=True)
        text = open(out).read()
        kernels, bad = scan(text)
        haz = scan_mfma_hazards(text)
        fly = scan_async_loads(text)
    print(f"{kernels} fused_fstat_kernel instantiations scanned, {len(bad)} early reads of in-flight asm ds_read destinations")
    print(f"{len(haz)} vector instructions reading an accumulator within 4 MFMAs of its last MFMA write")
    print(f"{len(fly)} readers of an in-flight asm feature load / AGPR shuffles")
    for b in bad[:10] + haz[:10] + fly[:10]:
        print("  ", b)
    return 1 if bad or haz or fly or kernels == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
