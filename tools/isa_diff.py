"""Which kernels' gfx950 ISA changed between a git revision and the working tree?

Development aid for work done without GPU time: a kernel whose instruction stream is identical to the one of a revision
that WAS measured cannot have changed its timing or its results; every kernel that differs is listed with its resource
usage (VGPRs, scratch, occupancy) before and after.  Usage:  python tools/isa_diff.py <rev> [file.hip ...]"""
import os
import re
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-value", "-Wno-unused-result",
         "-Rpass-analysis=kernel-resource-usage", "--save-temps", "-c"]


def compile_tree(tree, files, out):
    def one(f):
        d = os.path.join(out, f.replace(".hip", ""))
        os.makedirs(d, exist_ok=True)
        r = subprocess.run(["hipcc", *FLAGS, "-I" + os.path.join(tree, "include"), "-o", os.path.join(d, "o.o"),
                            os.path.join(tree, "theia_amd", "csrc", f)], cwd=d, capture_output=True, text=True)
        asm = [x for x in os.listdir(d) if x.endswith("gfx950.s")]
        return f, (os.path.join(d, asm[0]) if asm else None), r.stderr
    with ThreadPoolExecutor(8) as ex:
        return list(ex.map(one, files))


def kernels(asm_path, remarks):
    txt = open(asm_path).read()
    body = {}
    for f in re.split(r"\n(?=_Z\w+:)", txt):
        name = f.split(":", 1)[0]
        if not name.startswith("_Z") or ".amdhsa_kernel " + name not in txt:
            continue
        ins = [re.sub(r"\.LBB\d+_\d+", "L", l.strip()) for l in f.split("\n") if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        body[name] = ins
    res, cur = {}, None
    for line in remarks.split("\n"):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1); res[cur] = {}
            continue
        m = re.search(r"remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
        if m and cur:
            res[cur][m.group(1).split(" ")[0]] = m.group(2)
    return body, res


def main():
    rev = sys.argv[1]
    files = sys.argv[2:] or sorted(f for f in os.listdir(os.path.join(ROOT, "theia_amd", "csrc")) if f.endswith(".hip"))
    with tempfile.TemporaryDirectory() as tmp:
        old = os.path.join(tmp, "old"); os.makedirs(old)
        subprocess.run("git -C %s archive %s theia_amd/csrc include | tar -x -C %s" % (ROOT, rev, old), shell=True, check=True)
        files_old = [f for f in files if os.path.exists(os.path.join(old, "theia_amd", "csrc", f))]
        a = {f: (asm, err) for f, asm, err in compile_tree(old, files_old, os.path.join(tmp, "a"))}
        b = {f: (asm, err) for f, asm, err in compile_tree(ROOT, files, os.path.join(tmp, "b"))}
        changed = 0
        for f in files:
            if b[f][0] is None:
                sys.exit("compile failed: %s\n%s" % (f, b[f][1][-3000:]))
            def by_short(k, r):   # kernels are matched by name + template arguments: a new parameter changes the mangled name only
                out = {}
                for name in k:
                    short = subprocess.run(["c++filt", "-p", name], capture_output=True, text=True).stdout.strip() or name
                    out[short] = (k[name], r.get(name))
                return out
            kb = by_short(*kernels(*b[f]))
            ka = by_short(*kernels(*a[f])) if f in a and a[f][0] else {}
            for short in list(ka):   # a template parameter added with the default 'false' for the existing instantiations
                cand = short
                for _ in range(3):   # up to three added parameters
                    if cand in kb or not cand.endswith(">"):
                        break
                    cand = cand[:-1] + ", false>"
                if short not in kb and cand in kb:
                    ka[cand] = ka.pop(short)
                elif short not in kb and short + "<false>" in kb:
                    ka[short + "<false>"] = ka.pop(short)
            for short in sorted(set(ka) | set(kb)):
                if short not in ka:
                    print("%-22s NEW      %-60s %s" % (f, short[:60], kb[short][1])); continue
                if short not in kb:
                    print("%-22s REMOVED  %s" % (f, short[:60])); continue
                if ka[short][0] != kb[short][0]:
                    changed += 1
                    print("%-22s CHANGED  %-60s %d -> %d instructions; %s -> %s" % (f, short[:60], len(ka[short][0]), len(kb[short][0]), ka[short][1], kb[short][1]))
        print("kernels of %s whose ISA changed: %d (everything not listed is instruction-for-instruction identical)" % (rev, changed))


if __name__ == "__main__":
    main()
