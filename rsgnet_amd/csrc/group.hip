// Grouped launch: several independent convolutions and fuse-sums of one HighResolutionModule in ONE grid.
//
// The fuse layers of a module (pose_hrnet.py:190-268) are up to 12 small convolutions (1x1 at low
// resolution, strided 3x3 chains) plus one sum per output branch; as separate launches each is shorter
// than the cost of dispatching it.  They are grouped by dependency level instead: every conv whose
// inputs are ready and every sum whose addends are ready go into the same launch.  A workgroup finds its
// item from blockIdx.x and runs the matching instantiation of conv_body / fuse_body.
#include "conv_body.h"
#include "fuse_body.h"

#define RSG_GROUP_MAX_CONVS 12
#define RSG_GROUP_MAX_FUSES 4

struct GroupArgs {
    int n_conv, n_fuse;
    int first[RSG_GROUP_MAX_CONVS + RSG_GROUP_MAX_FUSES + 1];
    int variant[RSG_GROUP_MAX_CONVS];
    int nbx[RSG_GROUP_MAX_CONVS];          // blocks along x of conv i (its grid.y is folded into x)
    ConvArgs conv[RSG_GROUP_MAX_CONVS];
    FuseArgs fuse[RSG_GROUP_MAX_FUSES];
};

// variant id = ((cinb_idx * 3 + taps_idx) * 5 + cfg_idx); cinb in {32,64,48}, taps in {1x1,3x3,2x2 (cinb 32 only)},
// cfg in {(1,4,3),(1,4,1),(2,2,3),(2,2,1),(4,1,2)} = (WM,WN,NT).  cinb 48: the 48-channel maps of HRNet-W48.
static int variant_id(int cinb, int taps, int wm, int nt) {
    const int ci = cinb == 32 ? 0 : (cinb == 64 ? 1 : (cinb == 48 ? 2 : -1));
    const int ti = taps == 1 ? 0 : (taps == 9 ? 1 : (taps == 4 && cinb == 32 ? 2 : -1));
    int cf = -1;
    if (wm == 1 && nt == 3) cf = 0;
    else if (wm == 1 && nt == 1) cf = 1;
    else if (wm == 2 && nt == 3) cf = 2;
    else if (wm == 2 && nt == 1) cf = 3;
    else if (wm == 4 && nt == 2) cf = 4;
    if (ci < 0 || ti < 0 || cf < 0) return -1;
    return (ci * 3 + ti) * 5 + cf;
}

// The item table is the kernel argument (3.2 KB of the 4 KB kernarg segment): as a table in device memory every
// workgroup spent 3.5-5 k cycles on a chain of dependent scalar loads before it could start (stamp build: 0.4-0.5 k for
// the same conv launched alone with by-value arguments).
#ifdef RSG_CONV_STAMPS
extern "C" int rsg_conv_set_stamp_buffer(void* dev_ptr) {
    unsigned long long* p = (unsigned long long*)dev_ptr;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_conv_stamps), &p, sizeof(p)) != hipSuccess) RSG_FAIL("stamps: memcpy failed");
    return 0;
}
#endif

template <class DT>
__global__ void __launch_bounds__(256, 3) conv_group_kernel(const GroupArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    RSG_CSTAMP(0);
#ifdef RSG_CONV_STAMPS
    if (g_conv_stamps && threadIdx.x == 0) {
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)" : "=s"(xcc), "=s"(hwid));
        g_conv_stamps[(size_t)blockIdx.x * 8 + 7] = ((unsigned long long)xcc << 32) | hwid;
    }
#endif
    const int n = g.n_conv + g.n_fuse;
    // which item?  Constant indices, so the whole first[] table comes in with one batch of scalar loads (a while
    // loop made every step wait for its own load: 4-9 k cycles before the later items even started)
    int i = 0;
#pragma unroll
    for (int j = 1; j < RSG_GROUP_MAX_CONVS + RSG_GROUP_MAX_FUSES; ++j) i += (j < n && (int)blockIdx.x >= g.first[j]) ? 1 : 0;
    const int local = blockIdx.x - g.first[i];
    if (i >= g.n_conv) {
        fuse_body<DT>(g.fuse[i - g.n_conv], local, g.first[i + 1] - g.first[i]);
        return;
    }
    // one item's arguments into registers: a batch of scalar loads from the kernel-argument segment (a reference into
    // the by-value table makes hipcc copy the whole table to scratch)
    const ConvArgs a = g.conv[i];
    const int nbx = g.nbx[i];
    const int tile = rsg_xcd_tile(local, g.first[i + 1] - g.first[i]);      // an item's neighbouring tiles on one XCD
    const int bx = tile % nbx, by = tile / nbx;
#define RSG_V(ID, C, TH_, TW_, WM_, WN_, NT_) \
    case ID: conv_body<DT, C, TH_, TW_, WM_, WN_, NT_>(a, bx, by, smem); break;
#define RSG_V5(BASE, C, TH_, TW_)        \
    RSG_V(BASE + 0, C, TH_, TW_, 1, 4, 3) \
    RSG_V(BASE + 1, C, TH_, TW_, 1, 4, 1) \
    RSG_V(BASE + 2, C, TH_, TW_, 2, 2, 3) \
    RSG_V(BASE + 3, C, TH_, TW_, 2, 2, 1) \
    RSG_V(BASE + 4, C, TH_, TW_, 4, 1, 2)
    switch (g.variant[i]) {
        RSG_V5(0, 32, 1, 1)
        RSG_V5(5, 32, 3, 3)
        RSG_V5(10, 32, 2, 2)
        RSG_V5(15, 64, 1, 1)
        RSG_V5(20, 64, 3, 3)
        RSG_V5(30, 48, 1, 1)
        RSG_V5(35, 48, 3, 3)
    }
#undef RSG_V5
#undef RSG_V
}

extern "C" int rsg_conv_groupable(const rsg_conv_desc* d) {
    ConvArgs a;
    TilePlan t;
    dim3 grid;
    if (rsg_conv_prepare(d, a, t, grid)) return 0;
    if (grid.x > 32767) return 0;
    return variant_id(t.CINB, d->kh * d->kw, t.WM, t.NT) >= 0 ? 1 : 0;
}

extern "C" size_t rsg_group_table_bytes(void) { return sizeof(GroupArgs); }

// HOST: validates the items and writes the launch table (rsg_group_table_bytes) into table_host; the
// caller copies it to the device once.  launch2 = {grid blocks, dynamic LDS bytes}.
extern "C" int rsg_conv_fuse_group_plan(const rsg_conv_desc* convs, int32_t n_convs, const rsg_fuse_item* fuses,
                                        int32_t n_fuses, int32_t dtype, void* table_host, int32_t* launch2) {
    if (!table_host || !launch2) RSG_FAIL("group: null pointer");
    if (n_convs < 0 || n_convs > RSG_GROUP_MAX_CONVS || n_fuses < 0 || n_fuses > RSG_GROUP_MAX_FUSES ||
        n_convs + n_fuses < 1)
        RSG_FAIL("group: %d convs / %d fuses (max %d / %d)", n_convs, n_fuses, RSG_GROUP_MAX_CONVS, RSG_GROUP_MAX_FUSES);
    GroupArgs& g = *(GroupArgs*)table_host;
    memset(&g, 0, sizeof(g));
    g.n_conv = n_convs; g.n_fuse = n_fuses;
    int blocks = 0, lds = 0;
    for (int i = 0; i < n_convs; ++i) {
        TilePlan t;
        dim3 grid;
        if (convs[i].dtype != dtype) RSG_FAIL("group: mixed dtypes");
        if (rsg_conv_prepare(&convs[i], g.conv[i], t, grid)) return 1;
        const int v = variant_id(t.CINB, convs[i].kh * convs[i].kw, t.WM, t.NT);
        if (v < 0 || grid.x > 32767) RSG_FAIL("group: conv %d (CINB=%d k=%d WM=%d NT=%d) is not groupable", i, t.CINB, convs[i].kh, t.WM, t.NT);
        g.variant[i] = v;
        g.nbx[i] = (int)grid.x;
        g.first[i] = blocks;
        blocks += grid.x * grid.y;
        if (t.lds > lds) lds = t.lds;
    }
    for (int i = 0; i < n_fuses; ++i) {
        const rsg_fuse_item& f = fuses[i];
        int nb = 0;
        if (rsg_fuse_prepare(f.same, f.n_same, f.low, f.low_shift, f.n_low, f.out, f.batch, f.h, f.w, f.c, g.fuse[i], nb,
                             f.same_ld, f.low_ld))
            return 1;
        if (nb > 1024) nb = 1024;
        g.first[n_convs + i] = blocks;
        blocks += nb;
    }
    g.first[n_convs + n_fuses] = blocks;
    launch2[0] = blocks;
    launch2[1] = lds;
    return 0;
}

extern "C" int rsg_conv_fuse_group_run(const void* table_host, int32_t blocks, int32_t lds, int32_t dtype,
                                       void* stream) {
    if (!table_host || blocks < 1 || lds < 0 || lds > 64 * 1024) RSG_FAIL("group_run: bad arguments");
    static_assert(sizeof(GroupArgs) <= 4096, "the launch table travels as the kernel argument");
    const GroupArgs& g = *(const GroupArgs*)table_host;
    if (dtype == RSG_BF16)
        RSG_LAUNCH(conv_group_kernel<BF16>, dim3(blocks), dim3(256), lds, (hipStream_t)stream, g);
    else
        RSG_LAUNCH(conv_group_kernel<F16>, dim3(blocks), dim3(256), lds, (hipStream_t)stream, g);
    RSG_CHECK_LAUNCH("conv_fuse_group");
    return 0;
}
