"""rsg_fuse_module_plan (csrc/fuse_module.hip): host-side validation and job dealing -- no GPU needed.  The kernel itself is
checked on the GPU by tests/test_gpu_fuse_module.py."""
import ctypes as C

import numpy as np
import pytest

from rsgnet_amd import hip


def test_fuse_module_plan_rejects_what_the_kernel_does_not_cover():
    """Host-side validation: a module the band kernel cannot run must fail in _plan (the caller keeps the grouped path),
    never reach the launch."""
    lib = hip.lib()
    n = lib.rsg_fuse_module_table_bytes()
    tab = np.zeros(n, np.uint8)
    l2 = (C.c_int32 * 2)()

    def desc(nbr=3, h=64, w=48):
        d = hip.FuseModuleDesc()
        for j in range(nbr):
            d.x[j], d.x_ld[j] = 0x10000, 32 << j
        for i in range(2):
            d.y[i], d.y_ld[i] = 0x10000, 32 << i
        d.batch, d.h, d.w, d.n_branches, d.dtype = 2, h, w, nbr, hip.RSG_F16
        spec = [(1, 1, 32, hip.FM_UP_Y0), (2, 1, 32, hip.FM_UP_Y0), (2, 1, 64, hip.FM_UP_Y1), (0, 3, 64, hip.FM_DOWN_Y1)]
        d.n_convs = len(spec)
        for c, (src, k, cout, role) in enumerate(spec):
            e = d.convs[c]
            e.w_packed, e.bias, e.src, e.k, e.cout, e.role = 0x10000, 0x10000, src, k, cout, role
        return d

    assert lib.rsg_fuse_module_plan(C.byref(desc()), tab.ctypes.data, l2) == 0 and l2[0] == 2 * 4
    bad = desc(h=60)                                   # not a multiple of 8
    assert lib.rsg_fuse_module_plan(C.byref(bad), tab.ctypes.data, l2) == 1
    bad = desc(w=512)                                  # a band of it does not fit LDS
    assert lib.rsg_fuse_module_plan(C.byref(bad), tab.ctypes.data, l2) == 1
    bad = desc()
    bad.convs[1].role = hip.FM_UP_Y1                   # branch 2's addend of y0 is missing
    assert lib.rsg_fuse_module_plan(C.byref(bad), tab.ctypes.data, l2) == 1
    bad = desc()
    bad.convs[3].src = 2                               # a strided conv of branch 2
    assert lib.rsg_fuse_module_plan(C.byref(bad), tab.ctypes.data, l2) == 1
    bad = desc()
    bad.x_ld[1] = 32                                   # branch 1 has 64 channels
    assert lib.rsg_fuse_module_plan(C.byref(bad), tab.ctypes.data, l2) == 1
    assert lib.rsg_fuse_module_plan(C.byref(desc()), tab.ctypes.data, l2) == 0
    with pytest.raises(RuntimeError):
        hip.check(lib.rsg_fuse_module_run(tab.ctypes.data, 7, l2[1], hip.RSG_F16, None), "run")    # blocks != the table's


def test_fuse_module_jobs_cover_every_output_tile_once():
    """The dealt job lists: every (conv, 32-channel tile, 32-pixel group) of a band exactly once, phase A (the LDS-resident
    1x1 results) in front of every wave's sync job, one sync job per wave."""
    lib = hip.lib()
    for nbr, h, w, c0 in ((2, 64, 48, 32), (3, 64, 48, 32), (4, 64, 48, 32), (4, 96, 72, 32), (4, 96, 72, 48), (3, 96, 72, 48),
                          (2, 96, 72, 48), (4, 64, 48, 48)):
        r32 = lambda c: (c + 31) // 32 * 32
        d = hip.FuseModuleDesc()
        for j in range(nbr):
            d.x[j], d.x_ld[j] = 0x10000, r32(c0 << j)
        for i in range(2):
            d.y[i], d.y_ld[i] = 0x10000, r32(c0 << i)
        d.batch, d.h, d.w, d.n_branches, d.dtype, d.y1_fold, d.c0 = 2, h, w, nbr, hip.RSG_F16, int(nbr == 2), c0
        convs = []
        for i in range(nbr):
            for j in range(nbr):
                if j > i:
                    convs.append((j, 1, c0 << i, hip.FM_UP_Y0 if i == 0 else hip.FM_UP_Y1 if i == 1 else hip.FM_TO_GLOBAL))
                elif j < i and j <= 1:
                    convs.append((j, 3, (c0 << i) if i - j == 1 else (c0 << j), hip.FM_DOWN_Y1 if (i, j) == (1, 0) else hip.FM_TO_GLOBAL))
        d.n_convs = len(convs)
        for c, (src, k, cout, role) in enumerate(convs):
            e = d.convs[c]
            e.w_packed, e.bias, e.src, e.k, e.cout, e.role = 0x10000, 0x10000, src, k, cout, role
            if role == hip.FM_TO_GLOBAL:
                e.out16, e.out_ld = 0x10000, r32(cout)
        tab = np.zeros(lib.rsg_fuse_module_table_bytes(), np.uint8)
        l2 = (C.c_int32 * 2)()
        assert lib.rsg_fuse_module_plan(C.byref(d), tab.ctypes.data, l2) == 0, lib.rsg_last_error()
        info = (C.c_int32 * 4)()
        jobs = np.zeros((96, 4), np.uint8)
        first = (C.c_int32 * 17)()
        assert lib.rsg_fuse_module_table_jobs(tab.ctypes.data, info, first, jobs.ctypes.data) == 0
        R0 = info[0]
        assert R0 in (8, 16) and l2[0] == 2 * (h // R0) and 0 < l2[1] <= 160 * 1024
        seen = {}
        for wv in range(16):
            lst = [tuple(int(v) for v in jobs[k]) for k in range(first[wv], first[wv + 1])]
            syncs = [k for k, j in enumerate(lst) if j[0] == 255]
            assert len(syncs) == 1
            for k, (c, ct, g0, ng) in enumerate(lst):
                if c == 255:
                    continue
                lds_result = convs[c][3] in (hip.FM_UP_Y0, hip.FM_UP_Y1)
                assert (k < syncs[0]) == lds_result, (wv, lst)
                assert 1 <= ng <= (2 if convs[c][1] == 3 and convs[c][0] == 1 else 3)
                for g in range(g0, g0 + ng):
                    assert (c, ct, g) not in seen
                    seen[(c, ct, g)] = wv
        for c, (src, k, cout, role) in enumerate(convs):
            lvl = src + (1 if k == 3 else 0)
            groups = ((R0 >> lvl) * (w >> lvl) + 31) // 32
            for ct in range(r32(cout) // 32):
                for g in range(groups):
                    assert (c, ct, g) in seen, (nbr, c, ct, g)
        assert len(seen) == sum(((R0 >> (s + (k == 3))) * (w >> (s + (k == 3))) + 31) // 32 * (r32(co) // 32) for s, k, co, _ in convs)
