"""Development aid: the order of vector-memory loads / stores, `s_waitcnt vmcnt(n)` and MFMAs in a compiled kernel.

    python tools/audit_waits.py kernels_conv1x1.hip 'conv1x1_kernelILi4ELi1E'

compiles mica_amd/csrc/<file> for gfx950 with --save-temps into mica_amd/csrc/build/audit/ and prints, for every kernel whose mangled name
contains the pattern, the run-length-coded sequence  L = global/buffer load, S = store, W<n> = s_waitcnt vmcnt(n), M = v_mfma, | = s_barrier,
B = branch.  What to look for ("the order of the vector-memory queue"): vector-memory operations retire in order, so a `W0` in
front of the MFMAs of a software-pipelined loop means the prefetch the source code spells out does not exist in the binary - usually because the
loads sit behind a condition (the compiler may only leave outstanding what it can prove was issued) or are older than a load that is waited for."""
import glob, os, re, subprocess, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, pat = sys.argv[1], sys.argv[2]
out = os.path.join(root, "mica_amd", "csrc", "build", "audit")
os.makedirs(out, exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-w", "-c", src, "-o", os.path.join(out, "a.o"), "--save-temps=obj"]
               + sys.argv[3:], cwd=os.path.join(root, "mica_amd", "csrc"), check=True, stderr=subprocess.DEVNULL)
s = open(glob.glob(os.path.join(out, os.path.splitext(os.path.basename(src))[0] + "-hip-*gfx950*.s"))[0]).read()
for name in re.findall(r"^(\S*" + pat + r"\S*):", s, re.M):
    st = s.index(name + ":")
    seq = []
    for l in s[st:s.index("s_endpgm", st)].split("\n"):
        l = l.strip()
        if l.startswith(("global_load", "buffer_load")): seq.append("L")
        elif l.startswith(("global_store", "buffer_store")): seq.append("S")
        elif "vmcnt" in l: seq.append("W" + re.search(r"vmcnt\((\d+)\)", l).group(1))
        elif l.startswith("v_mfma"): seq.append("M")
        elif l.startswith("s_barrier"): seq.append("|")
        elif l.startswith("s_cbranch"): seq.append("B")
    runs, prev, n = [], None, 0
    for x in seq + [None]:
        if x == prev: n += 1
        else:
            if prev: runs.append(prev + (f"x{n}" if n > 1 else ""))
            prev, n = x, 1
    print(name[:90], "\n  ", " ".join(runs), "\n")
