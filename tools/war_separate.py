#!/usr/bin/env python3
"""DESIGN 9b', the one pattern no build had separated: a v_pk_fma_f32 whose 64-bit SOURCE register pair is overwritten by the
return of a ds_read issued within the next few instructions (write-after-read).  Reads the -O3 device assembly of
tools/trp_old_projection.hip and writes a copy with 16 wait states (2 x s_nop 7) in front of every such ds_read -- the packed
FMA has left the pipeline before the load that overwrites its operand is even issued; nothing else changes (same registers,
same loads, same order).  usage: war_separate.py in.s out.s [window]"""
import re
import sys


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def main(src, dst, window=3):
    lines = open(src).read().split("\n")
    out, hits, recent = [], 0, []          # recent: source register sets of the last `window` instructions that were packed FMAs
    for line in lines:
        s = line.strip()
        is_inst = bool(s) and not s.startswith((".", ";", "//")) and not s.endswith(":")
        if is_inst and s.startswith("ds_read"):
            ops = [t.strip() for t in s.split(None, 1)[1].split(",")]
            dest = regs(ops[0])
            if any(dest & r for r in recent):
                ind = re.match(r"\s*", line).group(0)
                out += [ind + "s_nop 7", ind + "s_nop 7"]
                hits += 1
        out.append(line)
        if is_inst:
            if s.startswith("v_pk_fma_f32"):
                ops = [t.strip() for t in s.split(None, 1)[1].split(",")]
                recent.append(set().union(*[regs(t.split()[0]) for t in ops[1:4]]))
            else:
                recent.append(set())
            recent = recent[-window:]
    open(dst, "w").write("\n".join(out))
    print(f"{hits} ds_read instructions overwrite a source of a v_pk_fma_f32 issued at most {window} instructions earlier: separated by 16 wait states")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 3)
