"""ISA audit of the hand-written kernels, from the device assembly of every csrc/*.hip:

  A. store-data hazard: a 12- / 16-byte buffer / global store whose data registers are rewritten by a vector instruction
     within the next two wait states (gfx950 reads the store data after issue; hipcc pads the hazard itself except for
     buffer stores whose scalar offset is a register -- conv3x3s.hip: store_b128_soff);
  B. kernels that mix LDS-DMA loads, asm waits and ordinary VGPR loads: for every ordinary load, the wait in front of the
     first reader of its destination, checked against the number of younger vector-memory operations issued in between
     (straight-line scan; a load whose first reader lies behind a branch is followed along the fall-through path and
     marked "path").

  python scripts/audit_isa.py [--build] [dir with *.s]      (default /tmp/isa_all; --build regenerates it with hipcc)
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "att-aspp-unet_amd", "csrc")
KERNEL_FILES = ["igemm", "igemm_group", "conv3x3", "conv3x3s", "wgrad", "wgrad3x3", "wgradL", "wgrad3x3r", "gate", "bn",
                "pointwise", "poolbranch"]


def build(out):
    os.makedirs(out, exist_ok=True)
    procs = []
    for f in KERNEL_FILES:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-I", os.path.join(ROOT, "include"),
               "-I", CSRC, "-Wno-unused-result", "-Wno-unused-value", "-S", "--cuda-device-only", "-o", os.path.join(out, f + ".s"),
               os.path.join(CSRC, f + ".hip")]
        procs.append(subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
    for p in procs:
        p.wait()


def regs(tok):
    """'v[98:101]' -> {98..101}, 'v62' -> {62}; anything else -> empty."""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def parse(path):
    """-> {kernel: [(lineno, mnemonic, [operands], in_asm)]} (labels as mnemonic '.label')."""
    kernels, cur, name, in_asm = {}, None, None, False
    for ln, line in enumerate(open(path), 1):
        s = line.split(";")[0].rstrip() if not line.lstrip().startswith(";;#") else line.strip()
        if line.lstrip().startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if line.lstrip().startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = m.group(1)
            cur = kernels.setdefault(name, [])
            continue
        if cur is None or not s.strip():
            continue
        t = s.strip()
        if t.startswith(".LBB"):
            cur.append((ln, ".label", [t.rstrip(":")], in_asm))
            continue
        if t.startswith(".") or t.startswith("s_endpgm"):
            if t.startswith("s_endpgm"):
                cur.append((ln, "s_endpgm", [], in_asm))
                cur = None
            continue
        parts = t.split(None, 1)
        ops = [o.strip() for o in re.split(r",\s*(?![^\[]*\])", parts[1])] if len(parts) > 1 else []
        cur.append((ln, parts[0], ops, in_asm))
    return kernels


VM_RE = re.compile(r"^(buffer_|global_|flat_|scratch_)(load|store|atomic)")


def is_vm(mn):
    return bool(VM_RE.match(mn))


def writes(mn, ops):
    """destination VGPRs of an instruction (first operand of vector ops and loads; permlane swaps write both)."""
    if mn.startswith(("v_cmp", "v_cmpx", "s_", "buffer_store", "global_store", "flat_store", "ds_write", "ds_store", ".label")):
        return set()
    if mn.startswith("v_permlane") and "swap" in mn:
        return regs(ops[0]) | regs(ops[1])
    if mn.startswith(("v_", "ds_read", "ds_bpermute", "ds_permute", "ds_swizzle")) or (is_vm(mn) and "load" in mn):
        if is_vm(mn) and ops and ops[-1].endswith("lds"):
            return set()
        return regs(ops[0]) if ops else set()
    return set()


def reads(mn, ops):
    r = set()
    start = 0 if mn.startswith(("buffer_store", "global_store", "flat_store", "ds_write", "ds_store", "v_cmp")) else 1
    for o in ops[start:]:
        r |= regs(o.split()[0]) if o else set()
    if mn.startswith("v_permlane") and "swap" in mn:
        r |= regs(ops[0])
    if mn.startswith(("v_mfma", "v_fmac", "v_mac", "v_dot2c", "v_pk_fmac")) and ops:
        r |= regs(ops[0]) if not mn.startswith("v_mfma") else set()
    return r


def wait_states(mn, ops):
    if mn == "s_nop":
        return int(ops[0]) + 1
    return 1


def audit_store_hazard(kern):
    out = []
    for i, (ln, mn, ops, in_asm) in enumerate(kern):
        if not re.match(r"^(buffer|global|flat)_store_dwordx[34]$", mn):
            continue
        data = regs(ops[0]) if mn.startswith("buffer") else regs(ops[1])
        ws, j = 0, i + 1
        while j < len(kern) and ws < 2:
            l2, m2, o2, _ = kern[j]
            if m2 == ".label":
                j += 1
                continue
            if m2.startswith("v_") and writes(m2, o2) & data:      # a VECTOR-ALU write (a later load's data returns far later)
                soff = ops[3].split()[0] if mn.startswith("buffer") and len(ops) > 3 else ""
                out.append((ln, mn, " ".join(ops), l2, m2, ws, soff))
                break
            ws += wait_states(m2, o2)
            j += 1
    return out


def audit_loads(kern):
    """B: ordinary VGPR loads in kernels that also use LDS-DMA and asm waits."""
    has_dma = any(is_vm(mn) and ops and ops[-1].endswith("lds") for _, mn, ops, _ in kern) or \
        any(mn.startswith("global_load_lds") for _, mn, _, _ in kern)
    has_asm_wait = any(mn == "s_waitcnt" and a for _, mn, _, a in kern)
    if not (has_dma and has_asm_wait):
        return None
    rows = []
    for i, (ln, mn, ops, in_asm) in enumerate(kern):
        if not (is_vm(mn) and "load" in mn) or (ops and ops[-1].endswith("lds")) or mn.startswith("global_load_lds"):
            continue
        dst = regs(ops[0])
        younger, best, path = 0, None, False
        verdict = "no reader found"
        for j in range(i + 1, min(len(kern), i + 4000)):
            l2, m2, o2, a2 = kern[j]
            if m2 == ".label":
                path = True
                continue
            if m2 in ("s_branch",):
                verdict = "path ends at a branch"
                break
            if m2 == "s_endpgm":
                break
            if m2 == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", " ".join(o2))
                if m and int(m.group(1)) <= younger:
                    best = (l2, int(m.group(1)), younger, "asm" if a2 else "hipcc")
            if reads(m2, o2) & dst:
                if best:
                    verdict = f"ok: vmcnt({best[1]}) [{best[3]}] at line {best[0]} with {best[2]} younger ops, first reader line {l2}"
                else:
                    verdict = f"VIOLATION: first reader {m2} at line {l2} with {younger} younger ops and no covering wait"
                break
            if writes(m2, o2) & dst and not (is_vm(m2)):
                verdict = f"dest rewritten at line {l2} before any reader (dead on this path)"
                break
            if is_vm(m2) or m2.startswith("global_load_lds"):
                younger += 1
        rows.append((ln, mn, ops[0], verdict + (" (path)" if path and verdict.startswith("ok") else "")))
    return rows


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    d = args[0] if args else "/tmp/isa_all"
    if "--build" in sys.argv or not os.path.isdir(d):
        build(d)
    nhaz = nviol = 0
    for f in sorted(os.listdir(d)):
        if not f.endswith(".s"):
            continue
        ks = parse(os.path.join(d, f))
        for name, kern in ks.items():
            short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
            hz = audit_store_hazard(kern)
            for ln, mn, ops, l2, m2, ws, soff in hz:
                nhaz += 1
                print(f"A {f}:{ln} {short}: {mn} {ops} -> data rewritten by {m2} (line {l2}) after {ws} wait state(s); soffset {soff!r}")
            rows = audit_loads(kern)
            if rows is None:
                continue
            bad = [r for r in rows if r[3].startswith("VIOLATION")]
            nviol += len(bad)
            nstore = sum(1 for _, mn, _, _ in kern if "store" in mn and is_vm(mn))
            print(f"B {f} {short}: {len(rows)} ordinary loads beside LDS-DMA + asm waits, {nstore} stores; "
                  f"{sum(1 for r in rows if r[3].startswith('ok'))} ok, {len(bad)} violations, "
                  f"{sum(1 for r in rows if not r[3].startswith(('ok', 'VIOLATION')))} other")
            for r in rows:
                if not r[3].startswith("ok"):
                    print(f"    line {r[0]} {r[1]} {r[2]}: {r[3]}")
    print(f"store-data hazards: {nhaz}; load-wait violations: {nviol}")


if __name__ == "__main__":
    main()
