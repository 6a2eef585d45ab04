#!/usr/bin/env python
"""Hand-edit the ISA of one kernel of a device assembly file (hipcc --cuda-device-only -S) and assemble the file into a
code object that the round-3 source tree's hunt hook launches instead of the compiled-in kernel (ISO_HSACO=...).
    python tools/isa_edit.py in.s out.hsaco KERNEL_SUBSTRING MODE [ARG]
MODE: none | hoist_exec (every `s_or_b64 exec, exec, sN` that follows a block label only after vector copies / SGPR moves is
      moved up to the label: the lanes a branch had switched off are back on BEFORE the copies run) | nop_after:<regex> (s_nop 7 after every instruction of the kernel that matches) | nop_before:<regex>
      | patch:<file> (lines "N<TAB>expected substring<TAB>replacement": line N of the kernel, counted from its label, is replaced)
The point: insert wait states around one class of instructions WITHOUT changing register allocation or instruction order."""
import re, subprocess, sys

LL = "/opt/rocm/lib/llvm/bin/"


def main():
    src, out, kern, mode = sys.argv[1:5]
    lines = open(src).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and kern in l and l.rstrip().endswith(":") or (l.startswith("_Z") and kern in l and ":" in l and "@" in l))
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    n = 0
    if mode.startswith("patch:"):
        for ln in open(mode[6:]):
            if not ln.strip() or ln.startswith("#"):
                continue
            num, expect, repl = ln.rstrip("\n").split("\t")
            i = start + int(num) - 1
            assert expect in lines[i], (num, expect, lines[i])
            lines[i] = "\t" + repl
            n += 1
    elif mode == "hoist_exec":
        i = start
        while i < end:
            if re.match(r"^\.LBB\d+_\d+:", lines[i].strip()):
                j = i + 1
                vec = 0
                while re.match(r"\s*(v_accvgpr_write_b32|scratch_store_dword\w*|s_mov_b64 s\[|s_mov_b32 s\d)", lines[j]):
                    vec += lines[j].strip().startswith(("v_", "scratch_"))
                    j += 1
                if vec and re.match(r"\s*s_or_b64 exec, exec, s\[", lines[j]):
                    print("hoisting line %d (%s) over %d lines after %s" % (j - start + 1, lines[j].strip().split(";")[0].strip(), j - i - 1, lines[i].strip()[:12]))
                    lines.insert(i + 1, lines.pop(j))
                    n += 1
            i += 1
    elif mode != "none":
        where, rx = mode.split(":", 1)
        pat = re.compile(rx)
        new = []
        for i, l in enumerate(lines):
            ins = l.strip()
            hit = start < i <= end and pat.search(ins) and not ins.startswith((";", "."))
            if hit and where == "nop_before":
                new.append("\ts_nop 7"); n += 1
            new.append(l)
            if hit and where == "nop_after":
                new.append("\ts_nop 7"); n += 1
        lines = new
    tmp = out + ".s"
    open(tmp, "w").write("\n".join(lines))
    obj = out + ".o"
    subprocess.check_call([LL + "clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", tmp, "-o", obj])
    subprocess.check_call([LL + "ld.lld", "-shared", obj, "-o", out])
    print("%s: kernel lines %d-%d, %d s_nop inserted" % (out, start, end, n))


if __name__ == "__main__":
    main()
