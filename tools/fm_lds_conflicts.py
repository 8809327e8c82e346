#!/usr/bin/env python3
"""LDS bank-conflict model of fuse_module_kernel's B-operand reads (csrc/fuse_module.hip): for every read pattern of a band
(1x1 convs at a branch's own resolution, 3x3 stride-2 convs of the de-interleaved images) the LDS cycles of one ds_read_b128 by
the lane groups of MI355X_MICROARCH.md (4 = conflict-free), and a search over the half offset / row pitch of a de-interleaved
image.  Host arithmetic only (no GPU): python tools/fm_lds_conflicts.py"""
# LDS bank-conflict model of fuse_module's B-operand reads (ds_read_b128): cycles per wave-instruction vs the 4-cycle ideal
G128 = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
G128 = G128 + [[l+32 for l in g] for g in G128]
def cycles(addrs):           # addrs: 64 byte addresses (16-B aligned); returns LDS cycles for ds_read_b128
    tot = 0
    for g in G128:
        banks = {}
        for l in g:
            a = addrs[l]
            if a is None: continue
            c = (a // 16) % 16
            banks.setdefault(c, set()).add(a)
        tot += max([len(v) for v in banks.values()] + [1])
    return tot
def fm_slot(x, hw): return ((x + 1) & 1) * hw + ((x + 1) >> 1)
def row_pitch(slots, ps, out_w):
    base = slots * ps; want = (out_w * ps // 2) % 128
    return base + ((want - base) % 128 + 128) % 128
def module(nbr, C0=32, W=48, R0=16, hw_override=None):
    halo = [0]*4; halo[0] = 1
    if nbr >= 3: halo[1] = 1
    res = {}
    for s in range(nbr):
        Wi = W >> s; ps = ((2*C0) << s) + 16
        rp = row_pitch(Wi + 1, ps, Wi // 2) if halo[s] else Wi * ps
        hw = (Wi + 2) >> 1
        if hw_override and s in hw_override:
            hw = hw_override[s]; rp = row_pitch(2*hw, ps, Wi//2) if halo[s] else rp
        # 1x1 reads at the source's own resolution (src >= 1)
        if s >= 1:
            OW, OR = Wi, R0 >> s; npx = OW*OR
            tot = n = 0
            for g in range((npx + 31)//32):
                for hh in (0,):
                    addrs = []
                    for lane in range(64):
                        nn_, h = lane & 31, lane >> 5
                        q = g*32 + nn_; qw = q if q < npx else q % npx
                        y, x = divmod(qw, OW)
                        slot = fm_slot(x, hw) if halo[s] else x
                        addrs.append((y + halo[s])*rp + slot*ps + h*16)
                    tot += cycles(addrs); n += 1
            res[f"1x1 from branch {s} (halo {halo[s]}, hw {hw}, rp {rp})"] = tot / n
        # 3x3 stride 2 reads (src 0, 1 where a conv exists)
        if s <= 1 and s + 1 < nbr and halo[s]:
            OW, OR = Wi // 2, (R0 >> s) // 2; npx = OW*OR
            for t in range(9):
                ky, kx = divmod(t, 3)
                tot = n = 0
                for g in range((npx + 31)//32):
                    addrs = []
                    for lane in range(64):
                        nn_, h = lane & 31, lane >> 5
                        q = g*32 + nn_; qw = q if q < npx else q % npx
                        y, x = divmod(qw, OW)
                        addrs.append((2*y)*rp + x*ps + h*16 + ky*rp + ((kx & 1)*hw + (kx >> 1))*ps)
                    tot += cycles(addrs); n += 1
                res.setdefault(f"3x3 s2 from branch {s} (hw {hw}, rp {rp})", []).append(tot / n)
    return res
for nbr in (2, 3, 4):
    print("module with", nbr, "branches")
    for k, v in module(nbr).items(): print("   ", k, v)
print("branch-1 half offset 25 slots:")
for k, v in module(4, hw_override={1: 25}).items(): print("   ", k, v)

def branch_reads(s, Wi, ps, hw, rp, R0s, halo=1):
    out = {}
    OW, OR = Wi, R0s; npx = OW*OR
    tot = n = 0
    for g in range((npx + 31)//32):
        addrs = []
        for lane in range(64):
            nn_, h = lane & 31, lane >> 5
            q = g*32 + nn_; qw = q if q < npx else q % npx
            y, x = divmod(qw, OW)
            addrs.append((y + halo)*rp + fm_slot(x, hw)*ps + h*16)
        tot += cycles(addrs); n += 1
    out['1x1'] = tot/n
    OW, OR = Wi // 2, R0s // 2; npx = OW*OR
    worst = 0
    for t in range(9):
        ky, kx = divmod(t, 3)
        tot = n = 0
        for g in range((npx + 31)//32):
            addrs = []
            for lane in range(64):
                nn_, h = lane & 31, lane >> 5
                q = g*32 + nn_; qw = q if q < npx else q % npx
                y, x = divmod(qw, OW)
                addrs.append((2*y)*rp + x*ps + h*16 + ky*rp + ((kx & 1)*hw + (kx >> 1))*ps)
            tot += cycles(addrs); n += 1
        worst = max(worst, tot/n)
    out['3x3'] = worst
    return out
print("search, branch 1 of HRNet-W32 (W 24, C 64, 8 rows):")
best = []
for hw in range(13, 30):
    for pad in range(0, 16):
        rp = (hw + 13) * 144 + pad * 16
        if hw + 13 < 25 and False: continue
        r = branch_reads(1, 24, 144, hw, rp, 8)
        best.append((r['1x1'] + r['3x3'], rp, hw, pad, r))
best.sort(key=lambda t: (t[0], t[1]))
for b in best[:8]: print(b)
print("W48 (C0 48: branch 1 W 36? no: 384x288 -> W0 72, branch 1 W 36, C 96, ps 208):")
best = []
for hw in range(19, 40):
    for pad in range(0, 16):
        rp = (hw + 19) * 208 + pad * 16
        r = branch_reads(1, 36, 208, hw, rp, 8)
        best.append((r['1x1'] + r['3x3'], rp, hw, pad, r))
best.sort(key=lambda t: (t[0], t[1]))
for b in best[:5]: print(b)
