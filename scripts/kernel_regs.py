"""Register / spill / LDS accounting of every gfx950 kernel in the built library.

Each object under ``urgent2026_challenge_track1_amd/build/`` carries its device code as a clang offload bundle in
``.hip_fatbin``; the bundle is unpacked with the ROCm LLVM tools and the AMDGPU metadata note of the code object
(``llvm-readelf --notes``) is read per kernel.  ``python scripts/kernel_regs.py`` writes
``profiles/r06_kernel_regs.json``; ``tests/test_kernel_regs.py`` asserts the shipping kernels spill nothing.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "urgent2026_challenge_track1_amd", "build")
LLVM = os.environ.get("ROCM_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
FIELDS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
          "private_segment_fixed_size", "group_segment_fixed_size", "max_flat_workgroup_size")


def _run(*cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("%s failed:\n%s" % (" ".join(cmd), r.stderr))
    return r.stdout


def demangle(names):
    if not names:
        return []
    r = subprocess.run(["c++filt"], input="\n".join(names) + "\n", capture_output=True, text=True)
    return r.stdout.splitlines() if r.returncode == 0 else list(names)


def object_kernels(obj):
    """[(mangled name, {field: int})] of one host object with an embedded gfx950 bundle."""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        co = os.path.join(tmp, "dev.co")
        try:
            _run(os.path.join(LLVM, "llvm-objcopy"), "--dump-section=.hip_fatbin=" + fat, obj)
        except RuntimeError as e:
            if "not found" in str(e):       # a host-only object (no kernels)
                return []
            raise
        if not os.path.exists(fat) or os.path.getsize(fat) == 0:
            return []
        _run(os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
             "--targets=" + TARGET, "--output=" + co)
        notes = _run(os.path.join(LLVM, "llvm-readelf"), "--notes", co)
        in_loop = scratch_in_loops(_run(os.path.join(LLVM, "llvm-objdump"), "-d", co))
    kernels = []
    cur = None
    for line in notes.splitlines():
        m = re.match(r"\s*(-\s+)?\.(\w+):\s+(\S+)\s*$", line)
        if not m:
            continue
        key, val = m.group(2), m.group(3)
        # a kernel's map starts at the list dash that precedes its first key (keys are sorted: .agpr_count first)
        if m.group(1) and key == "agpr_count":
            cur = {}
            kernels.append(cur)
        if cur is None:
            continue
        if key == "name":
            cur["name"] = val
        elif key in FIELDS:
            cur[key] = int(val)
    for k in kernels:
        if "name" in k:
            k["scratch_ops"], k["scratch_ops_in_loops"], k["scratch_ops_in_mfma_loops"] = in_loop.get(k["name"], (0, 0, 0))
    return [(k.pop("name"), k) for k in kernels if "name" in k]


def scratch_in_loops(disasm):
    """{kernel symbol: (scratch instructions, those inside a loop, those inside a loop that also holds MFMAs)} from `llvm-objdump -d` of a code
    object.  A spill that is stored and reloaded in a kernel's prologue costs nothing per iteration; one inside a time / K loop serialises the loads
    queued in front of it (the +5.7 ms of round 5).  Loop = [target, branch] of every backward branch of the function; the loops with
    matrix instructions are the hot ones of the GEMM / recurrence kernels (set-up loops - cluster formation spins, table fills - have none)."""
    out, sym, start, insts = {}, None, 0, []

    def close():
        if sym is None:
            return
        loops = [(t, a) for a, m, t in insts if t is not None and t <= a]
        sc = [a for a, m, t in insts if m.startswith("scratch_")]
        mf = [a for a, m, t in insts if m.startswith("v_mfma")]
        hot = [(lo, hi) for lo, hi in loops if any(lo <= a <= hi for a in mf)]
        out[sym] = (len(sc), sum(1 for a in sc if any(lo <= a <= hi for lo, hi in loops)),
                    sum(1 for a in sc if any(lo <= a <= hi for lo, hi in hot)))

    for line in disasm.splitlines():
        m = re.match(r"^([0-9a-f]{16}) <(\S+)>:", line)
        if m:
            close()
            sym, start, insts = m.group(2), int(m.group(1), 16), []
            continue
        m = re.match(r"^\s+(\S+)\s.*//\s*([0-9A-F]{12}):", line)
        if not m or sym is None:
            continue
        mn, addr = m.group(1), int(m.group(2), 16)
        tgt = None
        if mn.startswith("s_cbranch") or mn == "s_branch":
            t = re.search(r"<%s\+0x([0-9a-f]+)>" % re.escape(sym), line)
            tgt = start + int(t.group(1), 16) if t else (start if ("<%s>" % sym) in line else None)
        insts.append((addr, mn, tgt))
    close()
    return out


def short(name):
    """urse::kernel<args> without the parameter list."""
    depth = 0
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i].replace("void ", "")
    return name.replace("void ", "")


def collect():
    table = {}
    csrc = os.path.join(os.path.dirname(BUILD), "csrc")
    for f in sorted(os.listdir(BUILD)):
        if not f.endswith(".o") or not os.path.exists(os.path.join(csrc, f[:-2] + ".hip")):      # (an object whose source left the tree is not part of the library)
            continue
        ks = object_kernels(os.path.join(BUILD, f))
        names = demangle([n for n, _ in ks])
        for (mangled, info), dn in zip(ks, names):
            key = short(dn)
            info = dict(info, object=f)
            if key in table:          # same template instantiated in two objects: keep the worse one visible
                key = key + " @" + f
            table[key] = info
    return table


def main():
    sys.path.insert(0, ROOT)
    from urgent2026_challenge_track1_amd import build as _b
    _b.build_lib()
    table = collect()
    out = os.path.join(ROOT, "profiles", "r06_kernel_regs.json")
    if len(sys.argv) > 1:
        out = sys.argv[1]
    spilled = {k: v for k, v in table.items()
               if v.get("vgpr_spill_count", 0) or v.get("sgpr_spill_count", 0) or v.get("private_segment_fixed_size", 0)}
    with open(out, "w") as f:
        json.dump({"target": "gfx950", "n_kernels": len(table), "n_with_spills_or_scratch": len(spilled),
                   "with_spills_or_scratch": spilled, "kernels": table}, f, indent=1, sort_keys=True)
    print("%d kernels, %d with spills or scratch -> %s" % (len(table), len(spilled), out))
    for k, v in sorted(spilled.items()):
        print("  %-90s vgpr %3d spill v%d s%d scratch %d B" % (k[:90], v["vgpr_count"], v.get("vgpr_spill_count", 0),
                                                              v.get("sgpr_spill_count", 0), v.get("private_segment_fixed_size", 0)))


if __name__ == "__main__":
    main()
