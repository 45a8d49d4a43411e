// am_capi.hip -- host side of libairmodes_hip.so: the C ABI of include/airmodes_hip.h.
//
// Owns device buffers, the carry-over stream state (so that results do not depend on how the
// IQ stream is chunked) and the kernel sequence:
//   front end (IQ -> bb, avg)  ->  detect  ->  scan  ->  refine  ->  greedy chain (pointer
//   doubling)  ->  ordered compaction  ->  burst extraction  ->  slicer + CRC.
// There is no CPU fallback: every entry point that computes needs a HIP device.
#include "am_internal.h"

#include <chrono>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <thread>
#include <vector>

namespace {

// internal (never leaves this file): a speculative scan must be redone with the exact candidate count
#define AM_RETRY_EXACT 1000
// internal: the scan is enqueued, its completion is left to am_collect
#define AM_DEFERRED 1001

static inline double am_now_us()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

char g_create_err[256] = "";

} // namespace

struct am_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t own_stream = nullptr;   // created by am_create; `stream` is this one unless am_set_stream replaced it
    double rate = 0.0;
    uint64_t rate_i = 0;
    int spc = 0;                  // int(rate / 2e6): what the front end runs at (rx_path.py:35)
    int spc_hi = 0;               // samples per chip rounded UP: look-ahead sizes (= spc for a multiple of 2 MHz)
    bool frac = false;            // the rate is not a multiple of 2 MHz: rate-generic kernels, geometry from `geom`
    am_geom geom;                 // the preamble block's geometry in the reference's float arithmetic
    DevBuf chip_idx;              // [240] int: sample offset of soft chip j, int(j * samples per chip)
    float thr_db = 0.0f;
    float thr_lin = 0.0f;
    int use_pmf = 0;
    int tile = 0;
    // Speculative launches: the candidate count of a scan is only known on the device when its kernels
    // are enqueued.  Instead of a host round trip in the middle of the pipeline, the streaming path
    // launches for a capacity extrapolated from the previous scan (spec_cap) and lets the kernels clip
    // to the device-side count (Mdev); the real count comes back with the results, and a scan whose
    // count exceeded the capacity is redone with the exact count.
    bool allow_spec = true;       // (test builds: AIRMODES_NO_SPEC=1 disables)
    bool spec_now = false;        // this scan was launched for a capacity
    const uint32_t *Mdev = nullptr;
    double spec_density = 0.0;    // candidates per position, previous scan
    double spec_floor = 16384.0;  // slack added to the extrapolated capacity (AIRMODES_SPEC_FLOOR, tests)
    uint32_t ref_nseg = 0, ref_stride = 0, ref_endj = 0;   // arguments of the last run_refine (for the redo)
    int ref_mode = 0;
    const float *ref_bb = nullptr, *ref_avg = nullptr;
    int use_dcblock = 0;          // a2: dc_blocker_cc(100*spc, False) in front of |.|^2 (rx_path.py:39-41)
    const float *zt_base[2] = {nullptr, nullptr};   // where the zero tail of bb / avg was last written
    uint64_t zt_n[2] = {0, 0};
    // host-side wall-clock trace (AIRMODES_HOST_TRACE): cumulative microseconds per section
    double ht[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned ht_n = 0;
    uint32_t ticket_seq = 0;      // completion tickets (wait_for_ticket)
    bool dom_timed = false;       // ev[3] / ev[1] bracket the dominant kernel of this call
    // Every event record is a packet the in-order queue executes (a few microseconds each, measured in the
    // kernel trace).  ev[0] opens a call, ev[3] / ev[1] bracket the dominant kernel (ev[3] directly in front
    // of it: sharing ev[0] saved a packet but put the host's launch preparation into the kernel's time);
    // ev[2], the end of the device work, goes AFTER the completion ticket, i.e. off the path the host
    // waits on, and the whole-call time is formed on request (am_last_timing).
    bool total_pending = false;
    bool tail_synced = false;     // the stream is idle since the last scan's result synchronisation
    bool force_generic = false;   // (test builds: AIRMODES_GENERIC=1) use the rate-generic kernels only
    // am_submit_iq / am_collect: the scan of an independent batch is enqueued by one call and completed by the other,
    // so that one host thread can keep several contexts busy (am_pipe below)
    bool defer = false;           // set while am_submit_iq runs: chain_finish enqueues and returns AM_DEFERRED
    struct Pending {
        bool active = false;      // a submitted batch awaits am_collect
        bool scanned = false;     // ... and it has a scan in flight (a ticket to wait for)
        uint32_t seq = 0, M = 0, n_max = 0, cur0 = 0, emax = 0, max_hits = 0, j0 = 0, j1 = 0;
        const uint32_t *Mp = nullptr;
        uint64_t out_abs0 = 0, P1 = 0;
        double T0 = 0.0;
    } pend;
    bool keep_tags = false;       // AM_F_KEEP_TAGS of the call in progress: bursts + tags of its hits stay for am_fetch_tags
    uint64_t rec_base = 0;        // absolute index of array coordinate 0 of the resident records (am_fetch_candidates)
    bool poison = false;          // (test builds: AIRMODES_POISON=1) NaN-fill the sparse bb / reference-level arrays before every scan
    bool rows_in_gather = true;      // 64 Msps: bb rows around candidates from IQ in am_k_gather_wg (test builds: AIRMODES_ROWS_FE=1 keeps the front end's)
    bool rows_from_iq = false;       // ... in force for the scan in flight
    bool rows_max = true;            // ... with a maximum per row for am_k_refine_late (test builds: AIRMODES_ROWS_MAX=0 keeps round 5's first form)
    bool fused_refine = true;        // 64 Msps (round 6): list + rows + refinement in one launch, the rows in LDS (am_k_refine_seg); test builds:
                                     // AIRMODES_FUSED_REFINE=0 keeps am_k_gather_wg<1> + am_k_refine_late
    DevBuf bbmax;
    am_rows_args rows = {};
    bool allow_stream = true;        // (test builds: AIRMODES_FE=2) keeps the tile kernel (am_k_fe2, dense bb) where the streaming one would run
    // the scan whose records are resident: bb exists only around candidates (streaming front end), so burst
    // extraction recomputes from these samples (they must stay valid until the scan's hits are sliced)
    bool bb_sparse = false;
    int last_fe = 0;              // which front end the last scan ran (am_last_frontend)
    const float *scan_src = nullptr;
    uint64_t scan_src_abs0 = 0, scan_src_abs1 = 0;
    char err[256] = "";

    // "rx_time" stream tags still able to stamp a future preamble, ascending offsets (am_set_rx_time);
    // tt_dev mirrors them on the device for the extraction kernel
    std::vector<am_time_tag> tt;
    DevBuf tt_dev;
    am_time_tag *pin_tt = nullptr;   // pinned staging copy of the K streams' tags (multi_begin)

    // stream state (absolute sample indices)
    // K independent streams in one scan (am_process_multi): where stream j starts in the scanned buffer, the last position it may
    // emit (-1: none), and the packets each one got; hand_out() sorts the scan's packets into streams while this is set
    std::vector<uint64_t> multi_off, multi_cnt;
    std::vector<int64_t> multi_em;
    // the block-level preamble as a STREAM (am_preamble_stream): items so far, first undecided position, where the greedy scan
    // resumes, and the tail of both inputs that the next call's decisions still read
    uint64_t pb_total = 0, pb_next = 0, pb_cur = 0, pb_carry_abs0 = 0, pb_carry_n = 0;
    DevBuf pb_bb, pb_avg;
    uint64_t total_in = 0;    // samples received so far
    uint64_t next_pos = 0;    // first position whose preamble test is still undecided
    uint64_t chain_cur = 0;   // position at which the greedy scan resumes
    uint64_t carry_abs0 = 0;
    uint64_t carry_n = 0;
    DevBuf carry, carry2;

    // work buffers (grow only)
    DevBuf src, bb, avg, cand_seg, inavg, blk_cnt, blk_off, pos, e, tgt, valid,
        jump, emit_idx, dcount, off_local, blk_tot2, blk_base2, energy, bits, seg_base,
        cblk_cnt, cblk_off, scalars, bursts, tags, packets, crc_pow, recs, cscratch, dc_m1, dc_y, wgmax;
    uint32_t fe_vspan = 0, fe_nv = 0;  // streaming front end of the resident scan: array coordinates per workgroup, workgroups
    uint32_t fe_lag = 0, fe_wbits = 32;  // ... its bitmap: positions behind (lag), per word
    uint32_t fe_nwg = 0, fe_wpw = 0, fe_nwords = 0;   // ... front-end workgroups, bitmap words per workgroup, words in all
    uint32_t fe_nlong = 0, fe_wps = 0;    // ... levelled segments (am_launch_fe3): workgroups with fe_wpw words (the others: fe_wps fewer); 0: all alike
    int fe_wgs_per_cu = 0;                // persistent front-end workgroups per CU (0: as many as fit; am_pipe: one fewer)
    DevBuf lb_dc, lb_mark;      // slots of the chained scans (am_chain_prefix): zero at allocation, tagged with lb_epoch
    uint32_t lb_epoch = 0;
    uint32_t tk_base[2] = {0, 0};     // value of the ticket counters scalars[10], [11] when the next launch on them starts (am_chain_place)

    // results of the last scan
    std::vector<am_packet> h_packets;   // am_slicer_work: every sliced burst, reserved[0] = accepted
    std::vector<am_tag> h_tags;         // block-level scans (am_preamble_work): one tag per hit
    uint32_t n_hits = 0;                // preamble hits of the last scan
    std::vector<float> h_bursts;
    std::vector<am_packet> pending;     // accepted packets not yet handed to the caller
    uint64_t last_tags = 0;
    uint32_t last_M = 0;
    uint32_t chain_M = 0;               // records (or capacity) chain_prepare ran for
    bool jump_ready = false;            // the refinement already wrote the chain's successor array (am_k_cand)
    const uint32_t *chain_Mp = nullptr; // device-side count when chain_M is a capacity

    // time-sharded mode: the chunk whose bb/avg are resident
    uint64_t shard_base = 0, shard_start = 0, shard_end = 0, shard_total = 0;
    bool shard_ready = false;
    bool shard_more = false;            // the resident chunk was scanned with AM_F_MORE: the stream goes on, no end-of-stream rule
    void *keep_dst = nullptr;           // am_shard_keep_tail: copied between a resolve step's slicing and its completion
    const void *keep_src = nullptr;
    uint64_t keep_bytes = 0;
    bool resolving_shard = false;       // set around chain_finish by the am_shard_resolve* calls
    DevBuf shard_exit;                  // device word: where the scan left this context's chunk in the last resolved step (0: none)
    const am_entry_src *entry_src = nullptr; // set around chain_finish: the scan's start position is composed on the device (time shards)
    const uint32_t *flag_src = nullptr; // ... and this device word is handed to the host with the completion ticket (pin_scalars[4])
    hipEvent_t walk_event = nullptr;    // ... recorded behind the block walk (am_spipe: the next chunk's resolve step waits for this, not for the slicing)
    const uint64_t *word_src = nullptr; // ... and this 64-bit one (pin_scalars[12..13]: where the scan left the chunk -- am_spipe's books)

    // pinned host memory the tail kernels write into directly
    am_packet *pin_packets = nullptr;
    am_tag *pin_tags = nullptr;
    uint32_t *pin_scalars = nullptr;
    uint32_t pin_cap = 0;
    am_shard_exit *pin_exit = nullptr;  // exit table of a time chunk, written by its kernel directly
    uint32_t pin_exit_cap = 0;

    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_wait = nullptr;     // am_wait_for_stream
    float last_total_ms = 0.0f, last_dom_ms = 0.0f;
};

namespace {

int fail(am_ctx *c, int code, const char *what, hipError_t rc = hipSuccess)
{
    if (c) {
        if (rc != hipSuccess)
            snprintf(c->err, sizeof(c->err), "%s: %s", what, hipGetErrorString(rc));
        else
            snprintf(c->err, sizeof(c->err), "%s", what);
    }
    return code;
}

#define HIPCHK(c, call)                                                      \
    do {                                                                     \
        hipError_t rc__ = (call);                                            \
        if (rc__ != hipSuccess) return fail((c), AM_EHIP, #call, rc__);      \
    } while (0)

int ensure(am_ctx *c, DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap && b.p) return AM_OK;
    if (b.p) { (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
    // grow with headroom (hipFree / hipMalloc synchronise the whole device): half as much again for the
    // per-candidate arrays, whose size follows the traffic; an eighth for the big sample arrays
    size_t want = bytes + (bytes < ((size_t)64 << 20) ? bytes / 2 + ((size_t)256 << 10) : bytes / 8);
    hipError_t rc = hipMalloc(&b.p, want);
    if (rc != hipSuccess) { b.p = nullptr; return fail(c, AM_ENOMEM, "hipMalloc", rc); }
    b.cap = want;
    return AM_OK;
}

#define ENSURE(c, buf, bytes)                              \
    do {                                                   \
        int rc__ = ensure((c), (buf), (bytes));            \
        if (rc__ != AM_OK) return rc__;                    \
    } while (0)

// slots of a chained scan for n workgroups: zeroed when (re)allocated (epoch 0 is never used)
int ensure_slots(am_ctx *c, DevBuf &b, size_t n)
{
    const size_t bytes = (n + 8) * sizeof(unsigned long long);
    if (bytes <= b.cap && b.p) return AM_OK;
    int rc = ensure(c, b, bytes);
    if (rc != AM_OK) return rc;
    if (hipMemsetAsync(b.p, 0, b.cap, c->stream) != hipSuccess) return fail(c, AM_EHIP, "hipMemsetAsync");
    return AM_OK;
}

// tag of the next chained-scan launch.  0 is what freshly zeroed slots hold, so it is never handed out; when the
// 32-bit counter wraps, the slots are zeroed again (a slot beyond the recent grids could otherwise still carry the same
// number from 2^32 launches ago)
uint32_t next_epoch(am_ctx *c)
{
    if (++c->lb_epoch == 0) {
        for (DevBuf *b : {&c->lb_dc, &c->lb_mark})
            if (b->p) (void)hipMemsetAsync(b->p, 0, b->cap, c->stream);
        c->lb_epoch = 1;
    }
    return c->lb_epoch;
}

// the small device-side scalar block of a context: [0] resume position, [1] hit flag, [4..5] time-shard entry / overflow flag,
// zero when it is allocated
int ensure_scalars(am_ctx *c)
{
    if (c->scalars.p) return AM_OK;
    int rc = ensure(c, c->scalars, 16 * sizeof(uint32_t));
    if (rc != AM_OK) return rc;
    if (hipMemsetAsync(c->scalars.p, 0, 16 * sizeof(uint32_t), c->stream) != hipSuccess) return fail(c, AM_EHIP, "hipMemsetAsync");
    return AM_OK;
}

// the device word that carries the scan position from one time-shard step to the next: zero when allocated
int ensure_shard_exit(am_ctx *c)
{
    if (c->shard_exit.p) return AM_OK;
    int rc = ensure(c, c->shard_exit, 2 * sizeof(uint64_t));
    if (rc != AM_OK) return rc;
    if (hipMemsetAsync(c->shard_exit.p, 0, 2 * sizeof(uint64_t), c->stream) != hipSuccess) return fail(c, AM_EHIP, "hipMemsetAsync");
    return AM_OK;
}

void release(DevBuf &b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

// x^i mod G(x), G = 0x1FFF409: the syndrome of a frame is the XOR of these over its set bits
void crc_powers(uint32_t *t, int n)
{
    uint32_t v = 1;
    for (int i = 0; i < n; i++) {
        t[i] = v;
        v <<= 1;
        if (v & 0x1000000u) v = (v ^ 0xFFF409u) & 0xFFFFFFu;
    }
}

// lib/preamble_impl.cc:56-63 and the places that use d_samples_per_chip (:150,158-162,185,192,205-208,212,220,237), with the
// reference's types: float d_samples_per_chip = channel_rate / d_chip_rate (float / int), float d_samples_per_symbol
am_geom geom_of(uint64_t rate_i, int *idx /* [AM_BURST] */)
{
    am_geom g;
    const float channel_rate = (float)(int)rate_i;
    const float spcf = channel_rate / (float)2000000;
    const float sps = spcf * 2;
    g.S = (int)spcf;
    g.hist0 = (int)sps - 1;                              // set_history(d_samples_per_symbol): that many items, one of them current
    g.o1 = (int)(2 * spcf); g.o2 = (int)(7 * spcf); g.o3 = (int)(9 * spcf);
    g.late_max = (int)ceilf(spcf);                       // how_late < spcf holds for how_late = 0 .. ceil(spcf) - 1
    g.za0 = (int)(1.5 * sps);                            // (double product, as written there)
    g.za1 = (int)floorf(3 * sps);                        // largest j with (float)j <= 3 * sps
    g.zb0 = (int)(5 * sps);
    g.zb1 = (int)floor(7.5 * sps);                       // largest j with (double)j <= 7.5 * sps
    const float Bf = 240 * spcf;
    // consume_each(i + 240 * spc) (:237): int + float, rounded to float, truncated; i counts from the start of the current
    // general_work() window.  A whole number wherever 240 * spcf is one (every rate the tests pin against the reference's
    // C++); elsewhere (2.1 Msps: 251.99998) the sum rounds to i + 252 for all i >= 4: the skip at a representative window
    // offset, as oracle/airmodes_oracle.c::geom_of
    g.B = (int)((float)1024 + Bf) - 1024;
    g.room = (int)ceilf(Bf);                             // smallest d with !((float)d < Bf)
    for (int j = 0; j < AM_BURST; j++) idx[j] = (int)(j * spcf);
    g.span = idx[AM_BURST - 1];
    return g;
}

int configure_rate(am_ctx *c, double rate)
{
    if (!(rate >= 2e6) || rate > 2e9) return fail(c, AM_EINVAL, "rate must be at least 2 MHz (one sample per chip)");
    if (rate != floor(rate)) return fail(c, AM_EINVAL, "rate must be a whole number of samples per second");   // rx_path.py:33 int(rate)
    const int spc = (int)(rate / 2e6);                    // rx_path.py:35: the front end's window lengths
    const int tile = am_fe_pick_tile(spc);
    if (tile <= 0) return fail(c, AM_EINVAL, "samples per chip too large for the LDS tile");
    int idx[AM_BURST];
    const uint64_t rate_i = (uint64_t)(int)(float)rate;   // preamble_impl.cc:60: int d_sample_rate
    const am_geom g = geom_of(rate_i, idx);
    if (g.S != spc) return fail(c, AM_EINVAL, "rate not representable as the reference represents it (float)");
    c->rate = rate;
    c->rate_i = rate_i;
    c->spc = spc;
    c->frac = (double)spc * 2e6 != rate;
    c->spc_hi = c->frac ? spc + 1 : spc;
    c->geom = g;
    c->tile = tile;
    if (int rc = ensure(c, c->chip_idx, sizeof(idx)); rc != AM_OK) return rc;
    if (hipMemcpy(c->chip_idx.p, idx, sizeof(idx), hipMemcpyHostToDevice) != hipSuccess) return fail(c, AM_EHIP, "hipMemcpy (chip offsets)");
    return AM_OK;
}

void reset_stream(am_ctx *c)
{
    c->pb_total = c->pb_next = c->pb_cur = c->pb_carry_abs0 = c->pb_carry_n = 0;
    c->total_in = 0;
    c->next_pos = 0;
    c->chain_cur = 0;
    c->carry_abs0 = 0;
    c->carry_n = 0;
    c->shard_ready = false;
    c->tt.clear();            // item offsets restart with the stream
    if (c->shard_exit.p) (void)hipMemsetAsync(c->shard_exit.p, 0, 2 * sizeof(uint64_t), c->stream);   // the scan starts at sample 0 again
}

// positions beyond the end of the data read zeros: every bb/avg array carries this pad
inline uint64_t zero_pad(int spc_hi) { return (uint64_t)260 * (uint64_t)spc_hi + 64; }

// Wait for the end of a scan: the last launch stores a ticket number into pinned host memory and the
// host polls that word.  Asking the runtime instead (hipStreamSynchronize, hipEventQuery) adds tens of
// microseconds per scan -- its completion path runs through a helper thread.  After 50 ms of polling
// the thread blocks in the runtime.
hipError_t wait_for_ticket(am_ctx *c, uint32_t seq)
{
    volatile uint32_t *word = c->pin_scalars + 8;
    const double t0 = am_now_us();
    unsigned spins = 0;
    while (*word != seq) {
        if ((++spins & 255u) == 0) {
            // a scan takes a fraction of a millisecond: spin that long, then give the core away between looks,
            // and after 50 ms block in the runtime
            const double waited = am_now_us() - t0;
            if (waited > 50000.0) return hipStreamSynchronize(c->stream);
            if (waited > 2000.0) std::this_thread::yield();
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return hipSuccess;
}

// The `pad` floats after the n valid ones of a bb/avg array read as zero.  The kernels never write there,
// so the fill is skipped when the same place was zeroed last time (a launch less per call).
int zero_tail(am_ctx *c, float *base, uint64_t n, uint64_t pad, const float **last_base, uint64_t *last_n)
{
    if (*last_base == base && *last_n == n) return AM_OK;
    HIPCHK(c, hipMemsetAsync(base + n, 0, pad * sizeof(float), c->stream));
    *last_base = base;
    *last_n = n;
    return AM_OK;
}
#define ZERO_TAIL(c, which, base, n, pad)                                                        \
    do {                                                                                          \
        int rc__ = zero_tail((c), (base), (n), (pad), &(c)->zt_base[which], &(c)->zt_n[which]);   \
        if (rc__ != AM_OK) return rc__;                                                           \
    } while (0)

// a2: when the DC blocker is on, the path runs on y = dcblock(x) instead of x.  `*src`/`*src_abs0`
// describe the raw samples present, [need0, src_abs1) is what the front end will read; on return
// they describe the filtered samples.  Raw history of am_dcblock_history() samples before need0
// (or back to sample 0) must be present.
int apply_dcblock(am_ctx *c, const float **src, uint64_t *src_abs0, uint64_t src_abs1, uint64_t need0)
{
    if (!c->use_dcblock || src_abs1 <= need0) return AM_OK;
    const uint64_t H = am_dcblock_history(c->spc);
    const uint64_t raw0 = need0 > H ? need0 - H : 0;
    if (*src_abs0 > raw0) return fail(c, AM_EINVAL, "internal: DC blocker history was not carried");
    const uint64_t yn = src_abs1 - need0;
    ENSURE(c, c->dc_y, yn * 2 * sizeof(float));
    ENSURE(c, c->dc_m1, (yn + H / 2 + 1) * 2 * sizeof(float));
    HIPCHK(c, am_launch_dcblock(*src, (long long)*src_abs0, (long long)src_abs1, (long long)need0, (long long)yn,
                                c->spc, (float *)c->dc_m1.p, (float *)c->dc_y.p, c->stream));
    *src = (const float *)c->dc_y.p;
    *src_abs0 = need0;
    return AM_OK;
}

int run_frontend(am_ctx *c, const float *src, uint64_t src_abs0, uint64_t src_abs1, uint64_t out_abs0,
                 uint64_t out_n, float *bb, float *avg)
{
    am_fe_args a;
    a.iq = src;
    a.src_abs0 = (long long)src_abs0;
    a.src_abs1 = (long long)src_abs1;
    a.out_abs0 = (long long)out_abs0;
    a.out_n = (long long)out_n;
    a.bb = bb;
    a.avg = avg;
    a.spc = c->spc;
    a.use_pmf = c->use_pmf;
    a.tile = c->tile;
    a.s1 = (float)(1.0 / (double)c->spc);                           // rx_path.py:49
    a.sL = (float)(1.0 / (double)(AM_CHIPS_AVG * c->spc));          // rx_path.py:54
    HIPCHK(c, am_launch_frontend(a, c->stream));
    return AM_OK;
}

// Which kernels a scan of this context runs: 3 = the streaming front ends (am_k_fe3 / am_k_fe4: bitmap + sparse rows), 1 = the
// rate-generic kernels (dense bb / avg), 2 = the tile kernel am_k_fe2 (dense bb) -- in TEST builds only (-DAM_WITH_TILE_KERNEL:
// tests/gpu_variants, tests/emu; round 5: the product library no longer carries it nor the split refinement behind it).
// dense_wanted: the caller wants bb / avg as whole arrays (block-level am_frontend_work).
static int scan_path(const am_ctx *c, bool dense_wanted)
{
    if (c->force_generic || c->frac) return 1;
    if (!dense_wanted && c->allow_stream && am_fe4_supported(c->spc)) return 3;
#if AM_WITH_TILE_KERNEL
    if (am_fe2_tile(c->spc)) return 2;
#endif
    return 1;
}

// Scan of the per-segment candidate counts and read-back of the total; then either the
// refinement kernel (generic path: candidates only) or the gather of the records the fused
// kernel already produced.  Leaves the flat records (pos, e, tgt, inavg, valid) on the device.
int run_refine(am_ctx *c, const float *bb, const float *avg, uint32_t nseg, uint32_t seg_stride, int mode,
               uint32_t *M_out, uint32_t end_j = 0xFFFFFFFFu, uint32_t spec_cap = 0)
{
    *M_out = 0;
    c->spec_now = false;
    c->Mdev = nullptr;
    c->jump_ready = false;
    if (nseg == 0) return AM_OK;
    if (int rcs = ensure_scalars(c); rcs != AM_OK) return rcs;
    const uint32_t *count_ptr = (const uint32_t *)c->blk_off.p + nseg;       // device-side total
    if (mode == 3) {
        // streaming front end: the flat list is laid out from the per-workgroup counts by the gather kernel itself, which also
        // leaves the total (blk_off[0]); only a scan that must know its count up front adds the counts first
        count_ptr = (const uint32_t *)c->blk_off.p;
        if (!(spec_cap && mode >= 2)) {
            ENSURE(c, c->seg_base, ((size_t)c->fe_nwg + 2) * sizeof(uint32_t));
            HIPCHK(c, am_launch_scan_u32((uint32_t *)c->blk_cnt.p, (uint32_t *)c->seg_base.p, c->fe_nwg, c->stream));
            count_ptr = (const uint32_t *)c->seg_base.p + c->fe_nwg;
        }
    } else
        HIPCHK(c, am_launch_scan_u32((uint32_t *)c->blk_cnt.p, (uint32_t *)c->blk_off.p, nseg, c->stream));
    c->ref_bb = bb; c->ref_avg = avg; c->ref_nseg = nseg; c->ref_stride = seg_stride; c->ref_mode = mode;
    c->ref_endj = end_j;
    uint32_t M = 0;
    const uint32_t *Mp = nullptr;
    if (spec_cap && mode >= 2) {
        M = spec_cap;                                        // capacity; the kernels clip to *Mp
        Mp = count_ptr;                                      // (streaming front end: written by the gather kernel below)
        c->spec_now = true;
        c->Mdev = Mp;
    } else {
        HIPCHK(c, hipMemcpyAsync(&M, count_ptr, sizeof(uint32_t), hipMemcpyDeviceToHost,
                                 c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    if (M) {
        ENSURE(c, c->pos, ((size_t)M + 1) * sizeof(uint32_t));
        ENSURE(c, c->e, ((size_t)M + 1) * sizeof(uint32_t));
        ENSURE(c, c->tgt, ((size_t)M + 1) * sizeof(uint32_t));
        ENSURE(c, c->inavg, ((size_t)M + 1) * sizeof(float));
        ENSURE(c, c->valid, (size_t)M + 1);
        if (mode == 3 && c->rows_from_iq && c->fused_refine) {
            // 64 Msps: flat positions, bb rows (in LDS), late-peak decisions, quiet zones, records and the chain's successors in ONE
            // launch, one workgroup per front-end workgroup (am_k_refine_seg)
            ENSURE(c, c->jump, ((size_t)M + 1) * sizeof(uint32_t));
#if defined(AM_TEST_KNOBS)
            if (getenv("AIRMODES_TRACE_SPEC")) fprintf(stderr, "airmodes: am_k_refine_seg, %u segments of %u words, capacity %u\n", c->fe_nwg, c->fe_wpw, M);
#endif
            HIPCHK(c, am_launch_refine_seg((uint32_t *)c->bits.p, (uint32_t *)c->blk_cnt.p, (const float *)c->wgmax.p, c->fe_nwg, c->fe_nlong,
                                           c->fe_wpw, c->fe_wps, c->fe_nwords, M, c->fe_lag, c->fe_wbits, c->fe_vspan, c->fe_nv, c->rows, avg, c->thr_lin, end_j,
                                           (uint32_t *)c->pos.p, (uint32_t *)c->e.p, (uint32_t *)c->tgt.p, (float *)c->inavg.p,
                                           (uint8_t *)c->valid.p, (uint32_t *)c->jump.p, (uint32_t *)c->blk_off.p, c->stream));
            c->jump_ready = true;
        } else if (mode == 3) {
            // streaming front end: candidates arrive as a bitmap; flat positions, then late-peak decisions, quiet zones,
            // records and the chain's successors in one launch (am_k_refine_late)
            HIPCHK(c, am_launch_gather_wg((uint32_t *)c->bits.p, (uint32_t *)c->blk_cnt.p, c->fe_nwg, c->fe_wpw, c->fe_nwords, M,
                                          c->fe_lag, c->fe_wbits, (uint32_t *)c->pos.p, (uint32_t *)c->blk_off.p, c->stream,
                                          c->rows_from_iq ? &c->rows : nullptr));
            ENSURE(c, c->jump, ((size_t)M + 1) * sizeof(uint32_t));
            HIPCHK(c, am_launch_refine_late(bb, avg, (uint32_t *)c->pos.p, M, c->spc, c->thr_lin, end_j, (uint32_t *)c->e.p,
                                            (uint32_t *)c->tgt.p, (float *)c->inavg.p, (uint8_t *)c->valid.p,
                                            (uint32_t *)c->jump.p, c->stream, Mp, (const float *)c->wgmax.p, c->fe_vspan,
                                            c->fe_nv, c->rows_from_iq ? c->rows.bb_max : nullptr));
            c->jump_ready = true;
        }
#if AM_WITH_TILE_KERNEL
        else if (mode >= 2) {
            // split refinement behind the tile kernel: positions -> energy per reachable position -> per-candidate test
            const uint32_t nb = (M + 2047u) / 2048u;
            const uint64_t ebound = std::min<uint64_t>((uint64_t)M * (uint64_t)(c->spc + 1), (uint64_t)M + 0xFFFFFFFFull);
            ENSURE(c, c->dcount, ((size_t)M + 1) * sizeof(uint32_t));
            ENSURE(c, c->off_local, ((size_t)M + 1) * sizeof(uint32_t));
            ENSURE(c, c->blk_tot2, ((size_t)nb + 1) * sizeof(uint32_t));
            ENSURE(c, c->blk_base2, ((size_t)nb + 2) * sizeof(uint32_t));
            ENSURE(c, c->energy, (size_t)(ebound + 2) * sizeof(double));
            HIPCHK(c, am_launch_gather_pos((uint32_t *)c->cand_seg.p, seg_stride, (uint32_t *)c->blk_off.p, nseg, M,
                                           c->spc, (uint32_t *)c->pos.p, (uint32_t *)c->dcount.p, c->stream, Mp));
            // compact index of each candidate's first energy: one chained scan of the counts (global offsets)
            if (int rc = ensure_slots(c, c->lb_dc, nb); rc != AM_OK) return rc;
            HIPCHK(c, am_launch_exscan_chain((uint32_t *)c->dcount.p, (uint32_t *)c->off_local.p, M,
                                             (unsigned long long *)c->lb_dc.p, next_epoch(c), (uint32_t *)c->blk_base2.p + nb,
                                             (uint32_t *)c->scalars.p + 9, (uint32_t *)c->scalars.p + 10, &c->tk_base[0],
                                             c->stream, Mp));
            HIPCHK(c, am_launch_energy(bb, (uint32_t *)c->pos.p, (uint32_t *)c->dcount.p, (uint32_t *)c->off_local.p,
                                       nullptr, M, c->spc, (double *)c->energy.p, c->stream, Mp));
            ENSURE(c, c->jump, ((size_t)M + 1) * sizeof(uint32_t));
            HIPCHK(c, am_launch_cand(bb, avg, (uint32_t *)c->pos.p, (uint32_t *)c->dcount.p,
                                     (uint32_t *)c->off_local.p, nullptr, (double *)c->energy.p, M,
                                     c->spc, c->thr_lin, end_j, (uint32_t *)c->e.p, (uint32_t *)c->tgt.p,
                                     (float *)c->inavg.p, (uint8_t *)c->valid.p, (uint32_t *)c->jump.p, c->stream, Mp));
            c->jump_ready = true;
        }
#endif
        else
            HIPCHK(c, am_launch_refine(bb, avg, c->geom, c->thr_lin, (uint32_t *)c->cand_seg.p, seg_stride,
                                       (uint32_t *)c->blk_off.p, nseg, M, (uint32_t *)c->pos.p,
                                       (uint32_t *)c->e.p, (uint32_t *)c->tgt.p, (float *)c->inavg.p,
                                       (uint8_t *)c->valid.p, c->stream));
    }
    *M_out = M;
    return AM_OK;
}

// Candidate detection + refinement over positions [j0, j1) of existing device arrays bb/avg
// (block-level entry point: the generic detection kernel).
int run_candidates(am_ctx *c, const float *bb, const float *avg, uint32_t j0, uint32_t j1, uint32_t *M_out)
{
    *M_out = 0;
    c->bb_sparse = false;
    if (j1 <= j0) return AM_OK;
    const uint32_t nblk = (uint32_t)(((uint64_t)(j1 - j0) + AM_DET_PER_BLOCK - 1) / AM_DET_PER_BLOCK);
    ENSURE(c, c->cand_seg, (size_t)nblk * AM_DET_PER_BLOCK * sizeof(uint32_t));
    ENSURE(c, c->blk_cnt, (size_t)nblk * sizeof(uint32_t));
    ENSURE(c, c->blk_off, ((size_t)nblk + 1) * sizeof(uint32_t));
    HIPCHK(c, am_launch_detect(bb, avg, j0, j1, c->geom, c->thr_lin, (uint32_t *)c->cand_seg.p,
                               (uint32_t *)c->blk_cnt.p, nblk, c->stream));
    return run_refine(c, bb, avg, nblk, AM_DET_PER_BLOCK, 0, M_out);
}

// IQ -> bb, avg and the refined candidate records for positions [j0, j1): the fused
// specialisation when this samples-per-chip has one, the generic kernel pair otherwise.
int run_front_and_candidates(am_ctx *c, const float *src, uint64_t src_abs0, uint64_t src_abs1, uint64_t out_abs0,
                             uint64_t out_n, float *bb, float *avg, uint32_t j0, uint32_t j1, uint32_t *M_out,
                             bool may_speculate = false)
{
    *M_out = 0;
    c->spec_now = false;
    c->Mdev = nullptr;
    c->bb_sparse = false;
    const int path = scan_path(c, avg != nullptr);           // (fractional samples per chip: the rate-generic kernels)
    HIPCHK(c, hipEventRecord(c->ev[3], c->stream));
    c->last_fe = path == 1 ? 1 : 2;
    if (path == 1) {
        if (!avg) return fail(c, AM_EINVAL, "internal: the rate-generic kernels need the dense reference-level array");
        int rc = run_frontend(c, src, src_abs0, src_abs1, out_abs0, out_n, bb, avg);
        if (rc != AM_OK) return rc;
        HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
        c->dom_timed = true;
        return run_candidates(c, bb, avg, j0, j1, M_out);
    }
    c->scan_src = src; c->scan_src_abs0 = src_abs0; c->scan_src_abs1 = src_abs1;
    if (path == 3) {
        // streaming kernel: candidate bitmap + per-(step, wave) counts; bb and the reference level only around candidates
        const unsigned ns = am_fe4_steps((long long)out_n, c->spc);
        const unsigned wps = am_fe4_words(c->spc) * am_fe4_waves(c->spc);     // bitmap words per step
        ENSURE(c, c->bits, ((size_t)ns * wps + 64) * sizeof(uint32_t));
        ENSURE(c, c->blk_cnt, ((size_t)ns + 8) * sizeof(uint32_t));           // candidates per front-end workgroup
        ENSURE(c, c->blk_off, 16 * sizeof(uint32_t));                          // [0]: their total (am_k_gather_wg)
        ENSURE(c, c->avg, (out_n + zero_pad(c->spc_hi)) * sizeof(float));
        ENSURE(c, c->wgmax, ((size_t)ns + 8) * sizeof(float));
        unsigned nsteps = 0, spw = 1, nlong = 0;
        if (c->poison) {
            // test aid (AIRMODES_POISON=1): whatever the sparse arrays are read for must have been written by this scan
            HIPCHK(c, hipMemsetAsync(bb, 0xFF, out_n * sizeof(float), c->stream));
            HIPCHK(c, hipMemsetAsync(c->avg.p, 0xFF, out_n * sizeof(float), c->stream));
        }
        // 64 Msps (a bitmap word = one 32-sample chip, lag 288): the bb rows around candidates are formed from the samples by
        // am_k_gather_wg, not written by the front end (~42 MB of stores per 64 M samples that the dominant kernel does not make)
        c->rows_from_iq = c->rows_in_gather && am_fe4_unit(c->spc) == 32 && am_fe4_lag(c->spc) == 288;
        c->rows.iq = c->rows_from_iq ? src : nullptr;
        c->rows.src_abs0 = (long long)src_abs0; c->rows.src_abs1 = (long long)src_abs1; c->rows.out_abs0 = (long long)out_abs0;
        c->rows.out_n = (long long)out_n; c->rows.bb_sparse = bb; c->rows.use_pmf = c->use_pmf;
        c->rows.bb_max = nullptr;
        if (c->rows_from_iq && c->rows_max && !c->fused_refine) {
            // one float per array chip: the largest bb of every row formed (am_k_refine_late: whole chips of a quiet zone)
            ENSURE(c, c->bbmax, ((size_t)(out_n / 32) + 64) * sizeof(float));
            c->rows.bb_max = (float *)c->bbmax.p;
            if (c->poison) HIPCHK(c, hipMemsetAsync(c->bbmax.p, 0xFF, ((size_t)(out_n / 32) + 64) * sizeof(float), c->stream));
        }
        c->rows.s1 = (float)(1.0 / (double)c->spc);
        HIPCHK(c, am_launch_fe4(c->spc, src, (long long)src_abs0, (long long)src_abs1, (long long)out_abs0, (long long)out_n,
                                c->rows_from_iq ? nullptr : bb,
                                (float *)c->avg.p, j0, j1, c->use_pmf, (float)(1.0 / (double)c->spc),
                                (float)(1.0 / (double)(AM_CHIPS_AVG * c->spc)), c->thr_lin, (uint32_t *)c->bits.p,
                                (uint32_t *)c->blk_cnt.p, (float *)c->wgmax.p, &nsteps, &spw, c->stream, c->fe_wgs_per_cu,
                                (c->rows_from_iq && c->fused_refine) ? &nlong : nullptr));      // (levelled segments: what am_k_refine_seg can place)
        c->fe_vspan = spw * am_fe4_tile(c->spc);
        c->fe_nv = (nsteps + spw - 1) / spw;
        c->fe_nlong = 0;
        c->fe_wps = wps;
        if (nlong && spw > 1) {
            // nlong workgroups of spw steps, then workgroups of spw - 1
            const unsigned rest = nsteps > nlong * spw ? nsteps - nlong * spw : 0u;
            c->fe_nv = nlong + (rest + (spw - 1) - 1) / (spw - 1);
            c->fe_nlong = nlong;
        }
        c->fe_lag = am_fe4_lag(c->spc);
        c->fe_wbits = am_fe4_unit(c->spc);
        c->fe_nwg = c->fe_nv;
        c->fe_wpw = spw * wps;
        c->fe_nwords = nsteps * wps;
        HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
        c->dom_timed = true;
        c->bb_sparse = true;
        c->last_fe = 3;
        const uint64_t endj3 = src_abs1 > out_abs0 ? src_abs1 - out_abs0 : 0;
        uint32_t cap3 = 0;
        if (may_speculate && c->allow_spec && c->spec_density > 0.0) {
            const double npos = (double)(j1 - j0);
            const double want = c->spec_density * npos * 1.25 + c->spec_floor;
            cap3 = (uint32_t)std::max<double>(1.0, std::min<double>(want, std::min<double>(npos, 4.0e9)));
        }
        return run_refine(c, bb, (const float *)c->avg.p, c->fe_nwg, 0, 3, M_out,
                          (uint32_t)std::min<uint64_t>(endj3, 0xFFFFFFFFull), cap3);
    }
#if AM_WITH_TILE_KERNEL
    const unsigned T2 = am_fe2_tile(c->spc);
    const unsigned ntiles = (unsigned)((out_n + T2 - 1) / T2);
    const size_t nslots = (size_t)ntiles * T2;
    ENSURE(c, c->cand_seg, nslots * sizeof(uint32_t));
    ENSURE(c, c->blk_cnt, ((size_t)ntiles + 8) * sizeof(uint32_t));
    ENSURE(c, c->blk_off, ((size_t)ntiles + 9) * sizeof(uint32_t));
    unsigned nt = 0, tl = 0;
    // the fused kernel stops after detection and leaves avg[] around the candidates (or all of it, when
    // the caller wants the dense array); the refinement runs as separate kernels
    float *avg_sparse = nullptr;
    if (!avg) {
        ENSURE(c, c->avg, (out_n + zero_pad(c->spc_hi)) * sizeof(float));
        avg_sparse = (float *)c->avg.p;
    }
    HIPCHK(c, am_launch_fe2(c->spc, src, (long long)src_abs0, (long long)src_abs1, (long long)out_abs0,
                            (long long)out_n, bb, avg, j0, j1, c->use_pmf, (float)(1.0 / (double)c->spc),
                            (float)(1.0 / (double)(AM_CHIPS_AVG * c->spc)), c->thr_lin, (uint32_t *)c->cand_seg.p,
                            avg_sparse, (uint32_t *)c->blk_cnt.p, &nt, &tl, c->stream));
    HIPCHK(c, hipEventRecord(c->ev[1], c->stream));
    c->dom_timed = true;
    const uint64_t endj = src_abs1 > out_abs0 ? src_abs1 - out_abs0 : 0;
    uint32_t spec_cap = 0;
    if (may_speculate && c->allow_spec && avg_sparse && c->spec_density > 0.0) {
        // (a capacity of 0 would mean "exact": at least one slot)
        const double npos = (double)(j1 - j0);
        const double want = c->spec_density * npos * 1.25 + c->spec_floor;
        spec_cap = (uint32_t)std::max<double>(1.0, std::min<double>(want, std::min<double>(npos, 4.0e9)));
    }
    return run_refine(c, bb, avg_sparse ? avg_sparse : avg, nt, tl, 2, M_out,
                      (uint32_t)std::min<uint64_t>(endj, 0xFFFFFFFFull), spec_cap);
#else
    return fail(c, AM_EINVAL, "internal: no kernel for this scan");
#endif
}

// Greedy chain, part 1 (independent of where the scan starts): successor array and per-block exits
// over the M flat records (M may be a capacity, with the device-side count in Mp).
int chain_prepare(am_ctx *c, uint32_t M, bool want_last, const uint32_t *Mp = nullptr)
{
    c->chain_M = M;
    c->chain_Mp = Mp;
    if (M == 0) return AM_OK;
    const size_t stride = (size_t)M + 1;
    ENSURE(c, c->jump, stride * sizeof(uint32_t));
    if (int rc = ensure_scalars(c); rc != AM_OK) return rc;
    ENSURE(c, c->cscratch, am_chain_scratch_bytes(M));
    HIPCHK(c, am_launch_chain_prepare((uint32_t *)c->pos.p, (uint32_t *)c->tgt.p, M, (uint32_t *)c->jump.p,
                                      (uint32_t *)c->cscratch.p, want_last ? 1 : 0, c->stream, Mp, c->jump_ready ? 1 : 0));
    return AM_OK;
}

// The part of chain_finish behind the completion ticket: counts, resume position, accepted packets.
int chain_collect(am_ctx *c, uint32_t M, const uint32_t *Mp, uint32_t n_max, bool keep_bursts, uint32_t *final_cur)
{
    c->tail_synced = true;
    if (c->pin_scalars[5]) {
        // a chained scan gave up waiting for a workgroup that never published (am_chain_prefix): nothing of this step is valid
        (void)hipMemsetAsync((uint32_t *)c->scalars.p + 9, 0, sizeof(uint32_t), c->stream);
        return fail(c, AM_EHIP, "a chained scan on the device timed out (a workgroup never published its count)");
    }
    if (Mp) {
        // launched for a capacity: now the real candidate count is known
        c->last_M = c->pin_scalars[2];
        if (c->pin_scalars[2] > M) return AM_RETRY_EXACT;    // capacity too small: results are incomplete
    }
    const uint32_t n_emit = c->pin_scalars[0];
    *final_cur = c->pin_scalars[1];
    if (n_emit > n_max) return fail(c, AM_EHIP, "internal: more hits than the spacing bound allows");
    c->n_hits = n_emit;
    if (!keep_bursts) {
        const double TC = am_now_us();
        for (uint32_t i = 0; i < n_emit; i++) {
            if (!c->pin_packets[i].reserved[0]) continue;      // rejected: only this flag was written
            c->pending.push_back(c->pin_packets[i]);
            c->pending.back().reserved[0] = 0;
        }
        c->ht[7] += am_now_us() - TC;
        if (!c->keep_tags) return AM_OK;
    }
    c->h_tags.assign(c->pin_tags, c->pin_tags + n_emit);
    if (n_emit) {
        c->h_bursts.resize((size_t)n_emit * AM_BURST);
        HIPCHK(c, hipMemcpyAsync(c->h_bursts.data(), c->bursts.p, (size_t)n_emit * AM_BURST * sizeof(float),
                                 hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return AM_OK;
}

// Greedy chain, part 2: mark the candidates the scan visits when it starts at cur0, then extract
// and slice the hits (e <= emit_max, first-stage position in [own_lo, own_hi)).
// keep_bursts (block-level scan): fills h_tags + h_bursts.  Otherwise (streaming / sharded scan) the tags
// stay on the device and the accepted packets go straight from pinned memory to `pending`.
int chain_finish(am_ctx *c, const float *bb, uint32_t cur0, uint32_t emit_max, uint64_t base_abs,
                 bool keep_bursts, uint32_t *final_cur, uint32_t max_hits, uint32_t own_lo = 0,
                 uint32_t own_hi = 0xFFFFFFFFu, long long e_off = 0)
{
    const uint32_t M = c->chain_M;
    const uint32_t *Mp = c->chain_Mp;
    c->h_packets.clear();
    c->h_tags.clear();
    c->h_bursts.clear();
    c->n_hits = 0;
    c->last_M = M;
    c->rec_base = base_abs;
    *final_cur = cur0;
    if (M == 0) return AM_OK;
    const uint32_t nb = (uint32_t)(((uint64_t)M + AM_DET_PER_BLOCK - 1) / AM_DET_PER_BLOCK);
    ENSURE(c, c->cblk_off, ((size_t)nb + 1) * sizeof(uint32_t));
    // Hits are at least 240*spc apart, so their number is bounded by the span of the candidates;
    // everything downstream is launched for that bound and reads the real count on the device.
    // Packets and tags land directly in pinned host memory: one synchronisation for the whole tail.
    const uint32_t n_max = max_hits < M ? max_hits : M;
    uint32_t *n_ptr = (uint32_t *)c->cblk_off.p + nb;
    ENSURE(c, c->emit_idx, (size_t)n_max * sizeof(uint4));     // one record per hit: {candidate, position, refined position, reference level}
    if (int rc = ensure_slots(c, c->lb_mark, nb); rc != AM_OK) return rc;
    // which candidates the scan visits, which of them are hits, and their ordered list -- one launch after the walk
    HIPCHK(c, am_launch_chain_visit((uint32_t *)c->pos.p, (uint32_t *)c->jump.p, M, cur0, (uint32_t *)c->cscratch.p,
                                    (uint8_t *)c->valid.p, (uint32_t *)c->e.p, (uint32_t *)c->tgt.p, emit_max, own_lo,
                                    own_hi, (uint4 *)c->emit_idx.p, n_ptr, (unsigned long long *)c->lb_mark.p,
                                    next_epoch(c), (uint32_t *)c->scalars.p + 11, &c->tk_base[1],
                                    (uint32_t *)c->scalars.p, emit_max == 0xFFFFFFFFu ? 1 : 0, c->stream, Mp,
                                    c->entry_src, (const float *)c->inavg.p, c->walk_event));
    const bool keep_dev = keep_bursts || c->keep_tags;       // the bursts and their tags leave the kernel
    if (keep_dev) ENSURE(c, c->bursts, (size_t)n_max * AM_BURST * sizeof(float));
    if (c->pin_cap < n_max) {
        if (c->pin_packets) (void)hipHostFree(c->pin_packets);
        if (c->pin_tags) (void)hipHostFree(c->pin_tags);
        c->pin_packets = nullptr; c->pin_tags = nullptr; c->pin_cap = 0;
        const size_t want = (size_t)n_max + n_max / 4 + 64;
        HIPCHK(c, hipHostMalloc((void **)&c->pin_packets, want * sizeof(am_packet), hipHostMallocCoherent | hipHostMallocMapped));
        HIPCHK(c, hipHostMalloc((void **)&c->pin_tags, want * sizeof(am_tag), hipHostMallocCoherent | hipHostMallocMapped));
        c->pin_cap = (uint32_t)want;
    }
    if (!c->pin_scalars) {
        HIPCHK(c, hipHostMalloc((void **)&c->pin_scalars, 16 * sizeof(uint32_t), hipHostMallocCoherent | hipHostMallocMapped));
        memset(c->pin_scalars, 0, 16 * sizeof(uint32_t));     // [0..2] results of the slice launch, [3..4] time shards, [5] chained-scan error, [8] completion ticket
    }
    // extraction + slicing in one launch; the bursts and their tags leave the kernel only for the block-level
    // caller (am_preamble_work), the accepted packets always land in pinned host memory
    c->pin_scalars[0] = 0;
    c->pin_scalars[1] = cur0;
    c->pin_scalars[2] = 0;
    c->pin_scalars[5] = 0;
    // (the slicing waves write the accepted packets straight to the pinned array.  Routing them through device memory
    // and one coalesced copy in the ticket kernel was tried: the extraction kernel did not get faster and the copy
    // added 13 us to the ticket.)
    if (c->bb_sparse && !keep_bursts)
        // bb exists only around the candidates: the 240 soft chips of a hit are recomputed from the scan's samples
        HIPCHK(c, am_launch_extract_slice_iq(c->scan_src, (long long)c->scan_src_abs0, (long long)c->scan_src_abs1,
                                             c->use_pmf, (float)(1.0 / (double)c->spc), (const float *)c->inavg.p, c->spc,
                                             (const uint4 *)c->emit_idx.p, n_ptr, n_max, (uint32_t *)c->pos.p,
                                             (uint32_t *)c->e.p, base_abs, c->rate_i, (const am_time_tag *)c->tt_dev.p,
                                             (uint32_t)c->tt.size(), keep_dev ? (float *)c->bursts.p : nullptr,
                                             keep_dev ? c->pin_tags : nullptr, (uint32_t *)c->crc_pow.p,
                                             c->pin_packets, (uint32_t *)c->scalars.p, c->pin_scalars, c->stream, Mp));
    else
    HIPCHK(c, am_launch_extract_slice(bb, (const float *)c->inavg.p, c->spc, c->frac ? (const int *)c->chip_idx.p : nullptr,
                                      c->geom.hist0, (const uint4 *)c->emit_idx.p, n_ptr, n_max,
                                      (uint32_t *)c->pos.p, (uint32_t *)c->e.p, base_abs, e_off, c->rate_i,
                                      (const am_time_tag *)c->tt_dev.p, (uint32_t)c->tt.size(),
                                      keep_dev ? (float *)c->bursts.p : nullptr,
                                      keep_dev ? c->pin_tags : nullptr, (uint32_t *)c->crc_pow.p, c->pin_packets,
                                      (uint32_t *)c->scalars.p, c->pin_scalars, c->stream, Mp));
    if (c->keep_bytes && c->resolving_shard)
        // time shards: the samples the next step needs in front of its chunk, kept while this step's are still in place
        // (am_shard_keep_tail; behind the extraction kernel, which still reads them; complete when the ticket is seen)
        HIPCHK(c, hipMemcpyAsync(c->keep_dst, c->keep_src, c->keep_bytes, hipMemcpyDeviceToDevice, c->stream));
    const uint32_t seq = ++c->ticket_seq;
    HIPCHK(c, am_launch_ticket(c->pin_scalars + 8, seq, c->stream, c->flag_src, c->flag_src ? c->pin_scalars + 4 : nullptr,
                               c->word_src, c->word_src ? reinterpret_cast<uint64_t *>(c->pin_scalars + 12) : nullptr));
    HIPCHK(c, hipEventRecord(c->ev[2], c->stream));          // end of the device work of this scan (behind the ticket)
    c->total_pending = true;
    if (c->defer && !keep_bursts) {
        c->pend.scanned = true;
        c->pend.seq = seq; c->pend.M = M; c->pend.Mp = Mp; c->pend.n_max = n_max;
        return AM_DEFERRED;
    }
    const double TS = am_now_us();
    HIPCHK(c, wait_for_ticket(c, seq));
    c->ht[5] += am_now_us() - TS;
    return chain_collect(c, M, Mp, n_max, keep_bursts, final_cur);
}

int run_chain_and_slice(am_ctx *c, const float *bb, const float *, uint32_t M, uint32_t cur0, uint32_t emit_max,
                        uint64_t base_abs, bool keep_bursts, uint32_t *final_cur, uint32_t max_hits)
{
    c->h_packets.clear();
    c->h_tags.clear();
    c->h_bursts.clear();
    c->n_hits = 0;
    c->last_M = M;
    *final_cur = cur0;
    int rc = chain_prepare(c, M, false, c->spec_now ? c->Mdev : nullptr);
    if (rc != AM_OK || M == 0) return rc;
    return chain_finish(c, bb, cur0, emit_max, base_abs, keep_bursts, final_cur, max_hits);
}

void collect_accepted(am_ctx *c)
{
    for (const am_packet &p : c->h_packets) {
        if (!p.reserved[0]) continue;
        am_packet q = p;
        q.reserved[0] = 0;
        c->pending.push_back(q);
    }
}

// The scan ran over K streams laid out one behind the other (am_multi_layout): a packet belongs to the stream whose items its
// preamble lies in; positions a stream of its own would not have emitted (end-of-buffer rule, preamble_impl.cc:150,212 -- here they
// see the zeros of the gap, there the scan ends) are dropped, which changes nothing for the others: a hit only ever suppresses LATER
// positions, and the gap is longer than anything it can reach.  The item count becomes the stream's own; the time stamp already is
// (one internal "rx_time" tag of 0 s at every stream's first item).
void sort_into_streams(am_ctx *c)
{
    const uint64_t h0 = (uint64_t)c->geom.hist0;
    const size_t K = c->multi_off.size();
    c->multi_cnt.assign(K, 0);
    size_t w = 0;
    for (size_t i = 0; i < c->pending.size(); i++) {
        am_packet p = c->pending[i];
        const uint64_t pos = p.sample - h0;                       // position of the preamble in the scanned buffer
        size_t j = (size_t)(std::upper_bound(c->multi_off.begin(), c->multi_off.end(), pos) - c->multi_off.begin());
        if (j == 0) continue;
        j--;
        const uint64_t e = pos - c->multi_off[j];
        if (c->multi_em[j] < 0 || e > (uint64_t)c->multi_em[j]) continue;
        p.sample = e + h0;
        c->pending[w++] = p;
        c->multi_cnt[j]++;
    }
    c->pending.resize(w);
    c->multi_off.clear();
    c->multi_em.clear();
}

int hand_out(am_ctx *c, am_packet *out, uint64_t cap, uint64_t *n_out)
{
    if (!c->multi_off.empty()) sort_into_streams(c);
    const uint64_t n = c->pending.size();
    if (n_out) *n_out = n;
    if (n > cap) return fail(c, AM_ECAPACITY, "packet array too small; call am_fetch_packets");
    if (n && !out) return fail(c, AM_EINVAL, "null packet array");
    if (n) memcpy(out, c->pending.data(), n * sizeof(am_packet));
    c->pending.clear();
    return AM_OK;
}

// end-of-stream limits for a stream of N samples (preamble_impl.cc:150,212), in stream-index
// coordinates: positions n <= *emit_max may still be emitted.  Returns false if none can.
bool flush_limits(const am_ctx *c, uint64_t N, uint64_t *emit_max)
{
    const am_geom &g = c->geom;
    const uint64_t S = (uint64_t)g.S;
    const uint64_t K = N + (uint64_t)g.hist0;         // items incl. the block's history
    if (K - K % S <= S) return false;
    const uint64_t ninputs = K - K % S - S;           // :150
    const uint64_t need = (uint64_t)g.room + (uint64_t)g.hist0;
    if (ninputs < need) return false;
    *emit_max = ninputs - need;                       // k = n + hist0;  !(ninputs - k < 240 * samples per chip)
    return true;
}

} // namespace

extern "C" {

uint32_t am_abi_version(void) { return AM_ABI_VERSION; }

// 1 in the test-only CPU build of these sources (tests/emu: "device memory" is host memory there), 0 in the product
int am_is_emulated(void)
{
#if defined(AM_HIP_EMULATION)
    return 1;
#else
    return 0;
#endif
}

am_ctx *am_create(int device, double rate, float threshold_db, int use_pmf, int use_dcblock, int *err)
{
    int code = AM_OK;
    am_ctx *c = nullptr;
    g_create_err[0] = 0;
    do {
        int ndev = 0;
        hipError_t rc = hipGetDeviceCount(&ndev);
        if (rc != hipSuccess || ndev <= 0) {
            snprintf(g_create_err, sizeof(g_create_err), "no HIP device: %s",
                     rc != hipSuccess ? hipGetErrorString(rc) : "device count is 0");
            code = AM_ENODEV;
            break;
        }
        c = new (std::nothrow) am_ctx();
        if (!c) { code = AM_ENOMEM; break; }
        if (device < 0) {
            if (hipGetDevice(&device) != hipSuccess) device = 0;
        }
        if (device >= ndev) {
            snprintf(g_create_err, sizeof(g_create_err), "device %d out of range (%d devices)", device, ndev);
            code = AM_ENODEV;
            break;
        }
        c->device = device;
        if ((rc = hipSetDevice(device)) != hipSuccess || (rc = hipStreamCreate(&c->own_stream)) != hipSuccess) {
            snprintf(g_create_err, sizeof(g_create_err), "device setup: %s", hipGetErrorString(rc));
            code = AM_EHIP;
            break;
        }
        c->stream = c->own_stream;
        // timing events only: no system-scope cache flush when they execute (results reach the host through
        // pinned memory written by the kernels themselves, and through explicit copies)
        for (int i = 0; i < 4; i++) (void)hipEventCreateWithFlags(&c->ev[i], hipEventDisableSystemFence);
        c->use_pmf = use_pmf ? 1 : 0;
        c->use_dcblock = use_dcblock ? 1 : 0;
#if defined(AM_TEST_KNOBS)
        {
            // TEST BUILDS ONLY (tests/emu, tests/gpu_variants: -DAM_TEST_KNOBS): which kernels run, NaN-filled work arrays,
            // capacity launches.  The product library reads nothing from the environment.
            const char *g = getenv("AIRMODES_GENERIC");
            c->force_generic = g && g[0] == '1';
            const char *fe = getenv("AIRMODES_FE");
            c->allow_stream = !(fe && fe[0] == '2');
            const char *rf = getenv("AIRMODES_ROWS_FE");
            c->rows_in_gather = !(rf && rf[0] == '1');
            const char *rm = getenv("AIRMODES_ROWS_MAX");
            c->rows_max = !(rm && rm[0] == '0');
            const char *fr = getenv("AIRMODES_FUSED_REFINE");
            c->fused_refine = fr ? fr[0] == '1' : c->fused_refine;
            const char *po = getenv("AIRMODES_POISON");
            c->poison = po && po[0] == '1';
            const char *sp = getenv("AIRMODES_NO_SPEC");
            c->allow_spec = !(sp && sp[0] == '1');
            if (const char *sf = getenv("AIRMODES_SPEC_FLOOR")) c->spec_floor = atof(sf);
        }
#endif
        if ((code = configure_rate(c, rate)) != AM_OK) {
            snprintf(g_create_err, sizeof(g_create_err), "%s", c->err);
            break;
        }
        c->thr_db = threshold_db;
        c->thr_lin = powf(10.0f, (float)((double)threshold_db / 20.0));   // preamble_impl.cc:67
        uint32_t pw[112];
        crc_powers(pw, 112);
        if ((code = ensure(c, c->crc_pow, sizeof(pw))) != AM_OK) break;
        if ((rc = hipMemcpy(c->crc_pow.p, pw, sizeof(pw), hipMemcpyHostToDevice)) != hipSuccess) {
            snprintf(g_create_err, sizeof(g_create_err), "hipMemcpy: %s", hipGetErrorString(rc));
            code = AM_EHIP;
            break;
        }
    } while (0);
    if (code != AM_OK && c) { am_destroy(c); c = nullptr; }
    if (err) *err = code;
    return c;
}

#if defined(AM_XPROF)
extern "C" int am_debug_xprof(unsigned long long *out, int reset);
#endif
void am_destroy(am_ctx *c)
{
#if defined(AM_XPROF)
    {   // tuning builds: phase clocks of the extraction kernel since the last context went away
        unsigned long long acc[8];
        if (c && am_debug_xprof(acc, 1) == 0)
            fprintf(stderr, "xprof head %llu stage %llu sums %llu slice %llu\n", acc[0], acc[1], acc[2], acc[3]);
    }
#endif
    if (!c) return;
#if defined(AM_TEST_KNOBS)
    if (getenv("AIRMODES_HOST_TRACE") && c->ht_n)
        fprintf(stderr, "airmodes host trace over %u calls (us/call): setup %.1f, front end + refinement enqueue %.1f, chain + tail incl. sync %.1f (of which waiting %.1f), timing + hand-over %.1f, whole call %.1f, event-not-ready %.0f; accepted packets out of pinned memory %.1f\n",
                c->ht_n, c->ht[0] / c->ht_n, c->ht[1] / c->ht_n, c->ht[2] / c->ht_n, c->ht[5] / c->ht_n, c->ht[3] / c->ht_n, c->ht[4] / c->ht_n, c->ht[6], c->ht[7] / c->ht_n);
#endif
    (void)hipSetDevice(c->device);
    DevBuf *all[] = {&c->pb_bb, &c->pb_avg, &c->bbmax, &c->carry, &c->carry2, &c->src, &c->bb, &c->avg, &c->cand_seg, &c->inavg, &c->dcount, &c->off_local, &c->blk_tot2, &c->blk_base2,
                     &c->energy, &c->bits, &c->seg_base, &c->blk_cnt, &c->blk_off,
                     &c->pos, &c->e, &c->tgt, &c->valid, &c->jump, &c->emit_idx,
                     &c->lb_dc, &c->lb_mark, &c->cblk_cnt, &c->cblk_off, &c->scalars, &c->bursts, &c->tags, &c->packets, &c->crc_pow,
                     &c->recs, &c->cscratch, &c->dc_m1, &c->dc_y, &c->tt_dev, &c->wgmax, &c->shard_exit, &c->chip_idx};
    for (DevBuf *b : all) release(*b);
    if (c->pin_packets) (void)hipHostFree(c->pin_packets);
    if (c->pin_tags) (void)hipHostFree(c->pin_tags);
    if (c->pin_scalars) (void)hipHostFree(c->pin_scalars);
    if (c->pin_exit) (void)hipHostFree(c->pin_exit);
    if (c->pin_tt) (void)hipHostFree(c->pin_tt);
    for (int i = 0; i < 4; i++) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    if (c->ev_wait) (void)hipEventDestroy(c->ev_wait);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

int am_set_rate(am_ctx *c, double rate)
{
    if (!c) return AM_EINVAL;
    int rc = configure_rate(c, rate);
    if (rc == AM_OK) reset_stream(c);
    return rc;
}

int am_set_threshold(am_ctx *c, float threshold_db)
{
    if (!c) return AM_EINVAL;
    c->thr_db = threshold_db;
    c->thr_lin = powf(10.0f, (float)((double)threshold_db / 20.0));
    return AM_OK;
}

// preamble_impl.cc:165-170 latches the newest "rx_time" tag of the stream; tag_to_timestamp (:100-137)
// stamps every preamble relative to it.  Here the tags are handed over explicitly (no tagged stream at
// a C boundary) and take effect exactly at their offset.
int am_set_rx_time(am_ctx *c, uint64_t offset, uint64_t secs, double frac)
{
    if (!c) return AM_EINVAL;
    if (!(frac == frac)) return fail(c, AM_EINVAL, "rx_time: fractional part is NaN");
    if (!c->tt.empty() && offset < c->tt.back().offset)
        return fail(c, AM_EINVAL, "rx_time: tag offsets must not go backwards");
    // tags no future preamble can refer to: all but the newest one at or before the scan position
    const uint64_t scanned = std::max<uint64_t>(c->chain_cur, c->pb_cur);     // (am_process_iq's stream or am_preamble_stream's)
    size_t keep_from = 0;
    for (size_t i = 0; i < c->tt.size(); i++)
        if (c->tt[i].offset <= scanned) keep_from = i;
    if (keep_from) c->tt.erase(c->tt.begin(), c->tt.begin() + (long)keep_from);
    const am_time_tag t = {offset, secs, frac};
    if (!c->tt.empty() && c->tt.back().offset == offset) c->tt.back() = t;    // tstamp_tags.back(): the later one wins
    else {
        if (c->tt.size() >= AM_MAX_TIME_TAGS) return fail(c, AM_EINVAL, "rx_time: too many tags pending");
        c->tt.push_back(t);
    }
    // the context is idle between calls: a plain copy, no ordering against the stream needed
    HIPCHK(c, hipSetDevice(c->device));
    ENSURE(c, c->tt_dev, AM_MAX_TIME_TAGS * sizeof(am_time_tag));
    HIPCHK(c, hipMemcpy(c->tt_dev.p, c->tt.data(), c->tt.size() * sizeof(am_time_tag), hipMemcpyHostToDevice));
    return AM_OK;
}

int am_set_stream(am_ctx *c, void *hip_stream)
{
    if (!c) return AM_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));               // nothing of ours is left on the old one
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return AM_OK;
}

// Order the context's stream behind everything enqueued so far on another stream of the same device (hip_stream = NULL:
// the legacy default stream, which is PyTorch's "current stream" unless the caller changed it): an event recorded
// there, waited for here; the host does not block.  For inputs produced by somebody else's stream (an RCCL receive).
int am_wait_for_stream(am_ctx *c, void *hip_stream)
{
    if (!c) return AM_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->ev_wait) HIPCHK(c, hipEventCreateWithFlags(&c->ev_wait, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(c->ev_wait, (hipStream_t)hip_stream));
    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_wait, 0));
    return AM_OK;
}

// The other way round: another stream of the same device (NULL: the legacy default stream) waits, on the device, for what
// the context has enqueued so far -- e.g. a collective that sends what the context's kernels are still writing.
int am_signal_stream(am_ctx *c, void *hip_stream)
{
    if (!c) return AM_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->ev_wait) HIPCHK(c, hipEventCreateWithFlags(&c->ev_wait, hipEventDisableTiming));
    HIPCHK(c, hipEventRecord(c->ev_wait, c->stream));
    HIPCHK(c, hipStreamWaitEvent((hipStream_t)hip_stream, c->ev_wait, 0));
    return AM_OK;
}

double am_get_rate(const am_ctx *c) { return c ? c->rate : 0.0; }
float am_get_threshold(const am_ctx *c) { return c ? c->thr_db : 0.0f; }
int am_get_pmf(const am_ctx *c) { return c ? c->use_pmf : 0; }

int am_reset(am_ctx *c)
{
    if (!c) return AM_EINVAL;
    reset_stream(c);
    c->pending.clear();
    c->multi_off.clear();
    c->multi_em.clear();
    return AM_OK;
}

// am_process_iq, or with `submit` its first half: everything is enqueued, the wait for the scan and what follows it
// (packets, stream state) is left to am_collect.  submit needs AM_F_FLUSH: batches in flight are independent streams.
static int process_iq_core(am_ctx *c, const float *iq, uint64_t n, uint32_t flags, am_packet *out, uint64_t cap,
                           uint64_t *n_out, bool submit)
{
    if (!c) return AM_EINVAL;
    if (n_out) *n_out = 0;
    if (n && !iq) return fail(c, AM_EINVAL, "null iq");
    if (c->pend.active) return fail(c, AM_EINVAL, "a submitted batch has not been collected (am_collect)");
    if (submit && !(flags & AM_F_FLUSH)) return fail(c, AM_EINVAL, "am_submit_iq needs AM_F_FLUSH (independent batches)");
    const double T0 = am_now_us();
    HIPCHK(c, hipSetDevice(c->device));
    c->pending.clear();
    c->last_tags = 0;
    const bool flush = (flags & AM_F_FLUSH) != 0;
    const bool dev_in = (flags & AM_F_DEVICE_IN) != 0;
    c->keep_tags = (flags & AM_F_KEEP_TAGS) != 0;
    c->h_tags.clear();
    c->h_bursts.clear();
    const uint64_t S = (uint64_t)c->spc;
    const uint64_t L = (uint64_t)AM_CHIPS_AVG * S;
    const uint64_t LH = L + S;
    if (c->carry_n + n > ((uint64_t)1 << 31)) return fail(c, AM_EINVAL, "chunk larger than 2^31 samples");
    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    c->total_pending = false;
    c->dom_timed = false;
    c->tail_synced = false;

    // 1. one contiguous device view of [src_abs0, S1): carried tail + new samples
    const float *src = nullptr;
    uint64_t src_abs0 = c->total_in;
    if (c->carry_n == 0 && dev_in) {
        src = iq;                                       // zero copy
    } else if (c->carry_n + n > 0) {
        ENSURE(c, c->src, (c->carry_n + n) * 2 * sizeof(float));
        float *d = (float *)c->src.p;
        if (c->carry_n) {
            HIPCHK(c, hipMemcpyAsync(d, c->carry.p, c->carry_n * 2 * sizeof(float), hipMemcpyDeviceToDevice,
                                     c->stream));
            src_abs0 = c->carry_abs0;
        }
        if (n)
            HIPCHK(c, hipMemcpyAsync(d + c->carry_n * 2, iq, n * 2 * sizeof(float),
                                     dev_in ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream));
        src = d;
    }
    const uint64_t S1 = c->total_in + n;

    // 2. positions that can be decided now
    const uint64_t P0 = c->next_pos;
    uint64_t P1 = P0;
    uint64_t emit_max_abs = ~(uint64_t)0;
    if (flush) {
        uint64_t em;
        if (flush_limits(c, S1, &em)) {
            emit_max_abs = em;
            if (em + 1 > P0) P1 = em + 1;
        }
    } else {
        // a hit decided now must also be a hit if the stream ended right here
        const uint64_t hold = (uint64_t)(AM_BURST + 4) * (uint64_t)c->spc_hi;
        if (S1 > hold && S1 - hold > P0) P1 = S1 - hold;
    }
    if (P1 > P0) {
        const uint64_t out_abs0 = (P0 / L) * L;
        const uint64_t out_n = S1 - out_abs0;
        const uint64_t need0 = out_abs0 > LH ? out_abs0 - LH : 0;
        const float *fsrc = src;
        uint64_t fsrc_abs0 = src_abs0;
        {
            int rc = apply_dcblock(c, &fsrc, &fsrc_abs0, S1, need0);
            if (rc != AM_OK) return rc;
        }
        if (fsrc_abs0 > need0) return fail(c, AM_EINVAL, "internal: stream history was not carried");
        const uint64_t pad = zero_pad(c->spc_hi);
        const bool generic = scan_path(c, false) == 1;
        ENSURE(c, c->bb, (out_n + pad) * sizeof(float));
        float *bb = (float *)c->bb.p, *avg = nullptr;
        ZERO_TAIL(c, 0, bb, out_n, pad);
        if (generic) {          // the fused kernel carries the reference level in the candidate records
            ENSURE(c, c->avg, (out_n + pad) * sizeof(float));
            avg = (float *)c->avg.p;
            ZERO_TAIL(c, 1, avg, out_n, pad);
        }
        const uint32_t j0 = (uint32_t)(P0 - out_abs0), j1 = (uint32_t)(P1 - out_abs0);
        uint32_t M = 0;
        const double T1 = am_now_us();
        int rc = run_front_and_candidates(c, fsrc, fsrc_abs0, S1, out_abs0, out_n, bb, avg, j0, j1, &M, true);
        if (rc != AM_OK) return rc;
        const double T2 = am_now_us();
        c->ht[0] += T1 - T0; c->ht[1] += T2 - T1;
        const uint32_t cur0 = c->chain_cur > out_abs0 ? (uint32_t)std::min<uint64_t>(c->chain_cur - out_abs0, 0xFFFFFFF0u) : 0u;
        const uint32_t emax = emit_max_abs == ~(uint64_t)0 ? 0xFFFFFFFFu : (uint32_t)(emit_max_abs - out_abs0);
        uint32_t fin = cur0;
        const uint32_t max_hits = (uint32_t)((P1 - P0 + S) / ((uint64_t)AM_BURST * S) + 2);
        c->defer = submit;
        rc = run_chain_and_slice(c, bb, avg, M, cur0, emax, out_abs0, false, &fin, max_hits);
        c->defer = false;
        c->ht[2] += am_now_us() - T2;
        if (rc == AM_DEFERRED) {
            c->pend.active = true;
            c->pend.cur0 = cur0; c->pend.emax = emax; c->pend.max_hits = max_hits; c->pend.j0 = j0; c->pend.j1 = j1;
            c->pend.out_abs0 = out_abs0; c->pend.P1 = P1; c->pend.T0 = T0;
            return AM_OK;
        }
        if (rc == AM_RETRY_EXACT) {
#if defined(AM_TEST_KNOBS)
            if (getenv("AIRMODES_TRACE_SPEC")) fprintf(stderr, "airmodes: capacity %u < %u candidates, scan redone\n", M, c->last_M);
#endif
            // more candidates than the capacity this scan was launched for: redo the refinement and
            // the chain with the exact count (the fused kernel's outputs are still in place)
            rc = run_refine(c, c->ref_bb, c->ref_avg, c->ref_nseg, c->ref_stride, c->ref_mode, &M, c->ref_endj, 0);
            if (rc != AM_OK) return rc;
            fin = cur0;
            rc = run_chain_and_slice(c, bb, avg, M, cur0, emax, out_abs0, false, &fin, max_hits);
        }
        if (rc != AM_OK) return rc;
        c->spec_density = (j1 > j0) ? (double)c->last_M / (double)(j1 - j0) : 0.0;
        c->last_tags = c->n_hits;
        if (out_abs0 + fin > c->chain_cur) c->chain_cur = out_abs0 + fin;
        c->next_pos = P1;
    }

    if (submit) {                                           // nothing was scanned (or no candidate): still a batch to collect
        c->pend.active = true;
        c->pend.scanned = false;
        c->pend.T0 = T0;
        return AM_OK;
    }

    // 3. stream state for the next call
    if (flush) {
        reset_stream(c);
    } else {
        const uint64_t blk = (c->next_pos / L) * L;
        const uint64_t hist = LH + (c->use_dcblock ? am_dcblock_history(c->spc) : 0);
        const uint64_t C0 = blk > hist ? blk - hist : 0;
        const uint64_t keep = S1 - C0;
        if (keep) {
            ENSURE(c, c->carry2, keep * 2 * sizeof(float));
            HIPCHK(c, hipMemcpyAsync(c->carry2.p, src + (C0 - src_abs0) * 2, keep * 2 * sizeof(float),
                                     hipMemcpyDeviceToDevice, c->stream));
            std::swap(c->carry, c->carry2);
            c->tail_synced = false;
        }
        c->carry_abs0 = C0;
        c->carry_n = keep;
        c->total_in = S1;
    }
    if (!c->tail_synced) {
        // something was enqueued after the scan's own synchronisation (or no scan ran): the input
        // buffer must not be reused by the caller before the device is done with it
        HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->total_pending = false;
    }
    const double T5 = am_now_us();
    // (after a scan ev[2] sits behind the completion ticket and may still be in flight: am_last_timing waits
    // for it when somebody asks for the whole-call time)
    if (!c->total_pending) (void)hipEventElapsedTime(&c->last_total_ms, c->ev[0], c->ev[2]);
    c->last_dom_ms = 0.0f;
    if (c->dom_timed && hipEventElapsedTime(&c->last_dom_ms, c->ev[3], c->ev[1]) != hipSuccess) c->ht[6] += 1.0;
    const int hrc = hand_out(c, out, cap, n_out);
    c->ht[3] += am_now_us() - T5; c->ht[4] += am_now_us() - T0; c->ht_n++;
    return hrc;
}

int am_process_iq(am_ctx *c, const float *iq, uint64_t n, uint32_t flags, am_packet *out, uint64_t cap,
                  uint64_t *n_out)
{
    return process_iq_core(c, iq, n, flags, out, cap, n_out, false);
}

// ---- K independent streams in ONE scan (the 2 / 20 Msps configurations off the launch floor) -----------------
// Every stream is what am_process_iq(..., AM_F_FLUSH) takes: a whole stream from item 0.  They lie in ONE buffer, stream j at
// sample offset[j], zeros between them; one scan of the whole buffer serves all of them -- the eight launches of a scan then carry
// K streams' worth of samples.  Why this is exact and not an approximation:
//  * offsets are multiples of 48 samples-per-chip, so every sum of the canonical order (blocks aligned to the absolute
//    sample index) has the operands it has in the stream alone, and x + 0 = x;
//  * the gap is longer than the reference level's window behind a stream's first item and than the reach of a hit behind a
//    stream's last one (late shifts, 240 chips of skipped positions, the filter's tail), and a position whose own filter window is
//    all zeros is never a candidate (in[i] > inavg[i] * threshold is strict, preamble_impl.cc:174);
//  * what a stream of its own would not emit at its end is dropped afterwards (sort_into_streams).
static uint64_t multi_gap(const am_ctx *c) { return (uint64_t)(AM_BURST + 4 + 48 + 16) * (uint64_t)c->spc_hi + (uint64_t)c->geom.hist0; }

int am_multi_layout(am_ctx *c, uint32_t k, const uint64_t *n, uint64_t *offset, uint64_t *total)
{
    if (!c) return AM_EINVAL;
    if (!k || !n || !offset) return fail(c, AM_EINVAL, "multi: no streams");
    const uint64_t A = 48ull * (uint64_t)c->spc, G = multi_gap(c);
    uint64_t at = 0;
    for (uint32_t j = 0; j < k; j++) {
        offset[j] = at;
        const uint64_t end = at + n[j] + G;
        at = (end + A - 1) / A * A;
        if (j + 1 == k && total) *total = offset[j] + n[j];
    }
    return AM_OK;
}

// what am_process_multi and am_submit_multi share: the layout, the zeros, the per-stream limits and the internal time tags
static int multi_begin(am_ctx *c, float *iq, uint32_t k, const uint64_t *n, uint32_t flags, uint64_t *total)
{
    if (!k || !n) return fail(c, AM_EINVAL, "multi: no streams");
    if (k > AM_MAX_TIME_TAGS) return fail(c, AM_EINVAL, "multi: too many streams");
    if (c->use_dcblock) return fail(c, AM_EINVAL, "multi: not with the DC blocker (its delay line outlasts the gaps)");
    if (c->pend.active) return fail(c, AM_EINVAL, "multi: a submitted batch is waiting for am_collect");
    HIPCHK(c, hipSetDevice(c->device));
    reset_stream(c);                                              // every stream starts at item 0; pending rx_time tags are dropped
    std::vector<uint64_t> off(k);
    int rc = am_multi_layout(c, k, n, off.data(), total);
    if (rc != AM_OK) return rc;
    if (*total >= ((uint64_t)1 << 31)) return fail(c, AM_EINVAL, "multi: more than 2^31 samples in one scan");
    if (*total && !iq) return fail(c, AM_EINVAL, "null input");
    if (flags & AM_F_ZERO_GAPS)
        for (uint32_t j = 0; j + 1 < k; j++) {
            const uint64_t a = off[j] + n[j], b = off[j + 1];
            if (flags & AM_F_DEVICE_IN) HIPCHK(c, hipMemsetAsync(iq + 2 * a, 0, (b - a) * 2 * sizeof(float), c->stream));
            else memset(iq + 2 * a, 0, (b - a) * 2 * sizeof(float));
        }
    std::vector<int64_t> em_of(k, -1);
    c->tt.clear();
    for (uint32_t j = 0; j < k; j++) {
        uint64_t em;
        if (flush_limits(c, n[j], &em)) em_of[j] = (int64_t)em;
        const am_time_tag t = {off[j], 0, 0.0};                   // item counts and time restart with every stream
        c->tt.push_back(t);
    }
    ENSURE(c, c->tt_dev, AM_MAX_TIME_TAGS * sizeof(am_time_tag));
    // The context is idle -- its last scan was collected -- so nothing on its stream still reads the table or the staging copy.
    // Not a plain hipMemcpy: that one waits for the OTHER contexts' scans in flight (measured: am_pipe_submit_multi took a whole
    // scan's time, 360-410 us, on the host: tools/gpu_kstream_host_share.py)
    if (!c->pin_tt) HIPCHK(c, hipHostMalloc((void **)&c->pin_tt, AM_MAX_TIME_TAGS * sizeof(am_time_tag), hipHostMallocDefault));
    memcpy(c->pin_tt, c->tt.data(), c->tt.size() * sizeof(am_time_tag));
    HIPCHK(c, hipMemcpyAsync(c->tt_dev.p, c->pin_tt, c->tt.size() * sizeof(am_time_tag), hipMemcpyHostToDevice, c->stream));
    // the layout is put in force LAST: a failure above leaves the context a plain receiver (a stale layout would make the
    // next am_process_iq sort its packets into streams that are not there)
    c->multi_off = off;
    c->multi_em = em_of;
    c->multi_cnt.assign(k, 0);
    return AM_OK;
}

int am_process_multi(am_ctx *c, float *iq, uint32_t k, const uint64_t *n, uint32_t flags, am_packet *out, uint64_t cap,
                     uint64_t *count, uint64_t *n_out)
{
    if (!c) return AM_EINVAL;
    if (n_out) *n_out = 0;
    uint64_t total = 0;
    int rc = multi_begin(c, iq, k, n, flags, &total);
    if (rc != AM_OK) return rc;
    rc = process_iq_core(c, iq, total, (flags & AM_F_DEVICE_IN) | AM_F_FLUSH, out, cap, n_out, false);
    if (count) for (uint32_t j = 0; j < k; j++) count[j] = j < c->multi_cnt.size() ? c->multi_cnt[j] : 0;
    if (rc != AM_OK && rc != AM_ECAPACITY) { c->multi_off.clear(); c->multi_em.clear(); }
    return rc;
}

// The two halves, as am_submit_iq / am_collect are the halves of am_process_iq: am_submit_multi enqueues the scan of the K streams
// and returns; am_collect (the same call as for a single stream) waits for it and hands the packets out stream by stream;
// am_multi_counts then says how many each stream got.  Two contexts used alternately by one host thread keep the GPU busy while
// the host copies the previous scan's packets.
int am_submit_multi(am_ctx *c, float *iq, uint32_t k, const uint64_t *n, uint32_t flags)
{
    if (!c) return AM_EINVAL;
    uint64_t total = 0;
    int rc = multi_begin(c, iq, k, n, flags, &total);
    if (rc != AM_OK) return rc;
    rc = process_iq_core(c, iq, total, (flags & AM_F_DEVICE_IN) | AM_F_FLUSH, nullptr, 0, nullptr, true);
    if (rc != AM_OK) { c->multi_off.clear(); c->multi_em.clear(); }
    return rc;
}

int am_multi_counts(am_ctx *c, uint64_t *count, uint32_t k)
{
    if (!c) return AM_EINVAL;
    if (c->pend.active) return fail(c, AM_EINVAL, "multi: the submitted scan has not been collected yet (am_collect)");
    if (!count || k != c->multi_cnt.size()) return fail(c, AM_EINVAL, "multi: the last collected scan had another number of streams");
    for (uint32_t j = 0; j < k; j++) count[j] = c->multi_cnt[j];
    return AM_OK;
}

int am_submit_iq(am_ctx *c, const float *iq, uint64_t n, uint32_t flags)
{
    return process_iq_core(c, iq, n, flags, nullptr, 0, nullptr, true);
}

int am_collect(am_ctx *c, am_packet *out, uint64_t cap, uint64_t *n_out)
{
    if (!c) return AM_EINVAL;
    if (n_out) *n_out = 0;
    if (!c->pend.active) return hand_out(c, out, cap, n_out);        // (also: what an AM_ECAPACITY left behind)
    HIPCHK(c, hipSetDevice(c->device));
    am_ctx::Pending &P = c->pend;
    if (P.scanned) {
        const double TS = am_now_us();
        HIPCHK(c, wait_for_ticket(c, P.seq));
        c->ht[5] += am_now_us() - TS;
        uint32_t fin = P.cur0, M = P.M;
        int rc = chain_collect(c, P.M, P.Mp, P.n_max, false, &fin);
        if (rc == AM_RETRY_EXACT) {
            // more candidates than the capacity the scan was launched for: redo it with the exact count, now
            rc = run_refine(c, c->ref_bb, c->ref_avg, c->ref_nseg, c->ref_stride, c->ref_mode, &M, c->ref_endj, 0);
            if (rc == AM_OK) {
                fin = P.cur0;
                rc = run_chain_and_slice(c, (const float *)c->bb.p, nullptr, M, P.cur0, P.emax, P.out_abs0, false, &fin,
                                         P.max_hits);
            }
        }
        if (rc != AM_OK) { P.active = false; P.scanned = false; reset_stream(c); c->multi_off.clear(); c->multi_em.clear(); return rc; }
        c->spec_density = (P.j1 > P.j0) ? (double)c->last_M / (double)(P.j1 - P.j0) : 0.0;
        c->last_tags = c->n_hits;
    }
    else
        // nothing was scanned (a batch shorter than a burst): the copy of the caller's samples may still be in flight, and
        // the contract says they must stay valid only until am_collect returns
        HIPCHK(c, hipStreamSynchronize(c->stream));
    P.active = false;
    P.scanned = false;
    reset_stream(c);                                                  // (submitted batches end their stream: AM_F_FLUSH)
    c->last_dom_ms = 0.0f;
    if (c->dom_timed && hipEventElapsedTime(&c->last_dom_ms, c->ev[3], c->ev[1]) != hipSuccess) c->ht[6] += 1.0;
    c->ht[4] += am_now_us() - P.T0; c->ht_n++;
    return hand_out(c, out, cap, n_out);
}

int am_fetch_packets(am_ctx *c, am_packet *out, uint64_t cap, uint64_t *n_out)
{
    if (!c) return AM_EINVAL;
    return hand_out(c, out, cap, n_out);
}

uint64_t am_last_num_tags(const am_ctx *c) { return c ? c->last_tags : 0; }

int am_fetch_tags(am_ctx *c, float *bursts, am_tag *tags, uint64_t cap, uint64_t *n_out)
{
    if (!c) return AM_EINVAL;
    const uint64_t nt = c->h_tags.size();
    if (n_out) *n_out = nt;
    if (nt > cap) return fail(c, AM_ECAPACITY, "burst/tag arrays too small");
    if (nt && bursts) memcpy(bursts, c->h_bursts.data(), nt * AM_BURST * sizeof(float));
    if (nt && tags) memcpy(tags, c->h_tags.data(), nt * sizeof(am_tag));
    return AM_OK;
}

int am_fetch_candidates(am_ctx *c, uint64_t *pos, uint64_t *refined, uint8_t *valid, float *inavg, uint64_t cap,
                        uint64_t *n_out)
{
    if (!c) return AM_EINVAL;
    const uint64_t M = c->last_M;
    if (n_out) *n_out = M;
    if (M > cap) return fail(c, AM_ECAPACITY, "candidate arrays too small");
    if (M == 0) return AM_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    std::vector<uint32_t> t(M);
    if (pos) {
        HIPCHK(c, hipMemcpy(t.data(), c->pos.p, M * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < M; i++) pos[i] = c->rec_base + t[i];
    }
    if (refined) {
        HIPCHK(c, hipMemcpy(t.data(), c->e.p, M * sizeof(uint32_t), hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < M; i++) refined[i] = c->rec_base + t[i];
    }
    if (valid) HIPCHK(c, hipMemcpy(valid, c->valid.p, M, hipMemcpyDeviceToHost));
    if (inavg) HIPCHK(c, hipMemcpy(inavg, c->inavg.p, M * sizeof(float), hipMemcpyDeviceToHost));
    return AM_OK;
}

int am_frontend_work(am_ctx *c, const float *iq, uint64_t n, uint32_t flags, float *bb, float *avg)
{
    if (!c || (n && (!iq || !bb || !avg))) return fail(c, AM_EINVAL, "null argument");
    if (n == 0) return AM_OK;
    if (n > ((uint64_t)1 << 31)) return fail(c, AM_EINVAL, "stream larger than 2^31 samples");
    HIPCHK(c, hipSetDevice(c->device));
    const float *src = iq;
    if (!(flags & AM_F_DEVICE_IN)) {
        ENSURE(c, c->src, n * 2 * sizeof(float));
        HIPCHK(c, hipMemcpyAsync(c->src.p, iq, n * 2 * sizeof(float), hipMemcpyHostToDevice, c->stream));
        src = (const float *)c->src.p;
    }
    float *dbb = bb, *davg = avg;
    if (!(flags & AM_F_DEVICE_OUT)) {
        ENSURE(c, c->bb, n * sizeof(float));
        ENSURE(c, c->avg, n * sizeof(float));
        dbb = (float *)c->bb.p;
        davg = (float *)c->avg.p;
    }
    uint64_t s0 = 0;
    int rc = apply_dcblock(c, &src, &s0, n, 0);
    if (rc != AM_OK) return rc;
    if (scan_path(c, true) == 2) {
        uint32_t M = 0;      // (test builds) the tile kernel with an empty detection range: bb/avg only
        rc = run_front_and_candidates(c, src, 0, n, 0, n, dbb, davg, 0, 0, &M);
    } else {
        rc = run_frontend(c, src, 0, n, 0, n, dbb, davg);
    }
    if (rc != AM_OK) return rc;
    if (!(flags & AM_F_DEVICE_OUT)) {
        HIPCHK(c, hipMemcpyAsync(bb, dbb, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(avg, davg, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return AM_OK;
}

int am_preamble_work(am_ctx *c, const float *in, const float *inavg, uint64_t n, uint32_t flags,
                     float *bursts, am_tag *tags, uint64_t cap, uint64_t *n_out)
{
    if (!c) return AM_EINVAL;
    if (n_out) *n_out = 0;
    if (n && (!in || !inavg)) return fail(c, AM_EINVAL, "null input");
    if (n == 0) return AM_OK;
    if (n > ((uint64_t)1 << 31)) return fail(c, AM_EINVAL, "stream larger than 2^31 items");
    HIPCHK(c, hipSetDevice(c->device));
    const uint64_t pad = zero_pad(c->spc_hi);
    ENSURE(c, c->bb, (n + pad) * sizeof(float));
    ENSURE(c, c->avg, (n + pad) * sizeof(float));
    float *bb = (float *)c->bb.p, *avg = (float *)c->avg.p;
    const hipMemcpyKind kind = (flags & AM_F_DEVICE_IN) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    HIPCHK(c, hipMemcpyAsync(bb, in, n * sizeof(float), kind, c->stream));
    HIPCHK(c, hipMemcpyAsync(avg, inavg, n * sizeof(float), kind, c->stream));
    ZERO_TAIL(c, 0, bb, n, pad);
    ZERO_TAIL(c, 1, avg, n, pad);
    uint64_t em = 0;
    c->h_packets.clear();
    c->h_tags.clear();
    c->h_bursts.clear();
    if (flush_limits(c, n, &em)) {
        uint32_t M = 0, fin = 0;
        int rc = run_candidates(c, bb, avg, 0, (uint32_t)(em + 1), &M);
        if (rc != AM_OK) return rc;
        rc = run_chain_and_slice(c, bb, avg, M, 0, (uint32_t)em, 0, true, &fin,
                                 (uint32_t)(n / ((uint64_t)AM_BURST * (uint64_t)c->spc) + 2));
        if (rc != AM_OK) return rc;
    }
    const uint64_t nt = c->h_tags.size();
    if (n_out) *n_out = nt;
    if (nt > cap) return fail(c, AM_ECAPACITY, "burst/tag arrays too small");
    if (nt) {
        if (!bursts || !tags) return fail(c, AM_EINVAL, "null output");
        memcpy(bursts, c->h_bursts.data(), nt * AM_BURST * sizeof(float));
        memcpy(tags, c->h_tags.data(), nt * sizeof(am_tag));
    }
    return AM_OK;
}

// The preamble block as the streaming gr::block it is in the reference (include/gr_air_modes/preamble.h:36-46,
// lib/preamble_impl.cc:139-246): call after call on consecutive pieces of the two input streams, the result is what ONE work()
// call over their concatenation gives -- positions are decided when (240 + 4) samples-per-chip items of look-ahead
// exist, the undecided tail of both inputs is carried to the next call, the greedy scan resumes where it stopped
// (consume_each, :213,237,244), item counts and time stamps keep counting; AM_F_FLUSH ends the stream under the end-of-buffer rule
// (:150,212) and starts a new one at item 0.  am_reset() drops the state; am_set_rx_time offsets are stream-absolute.
int am_preamble_stream(am_ctx *c, const float *in, const float *inavg, uint64_t n, uint32_t flags,
                       float *bursts, am_tag *tags, uint64_t cap, uint64_t *n_out)
{
    if (!c) return AM_EINVAL;
    if (n_out) *n_out = 0;
    if (n && (!in || !inavg)) return fail(c, AM_EINVAL, "null input");
    if (c->pb_carry_n + n > ((uint64_t)1 << 31)) return fail(c, AM_EINVAL, "piece larger than 2^31 items");
    HIPCHK(c, hipSetDevice(c->device));
    const bool flush = (flags & AM_F_FLUSH) != 0;
    const hipMemcpyKind kind = (flags & AM_F_DEVICE_IN) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    const uint64_t pad = zero_pad(c->spc_hi);
    // one contiguous device view of the items [A0, S1): carried tail + new ones; array coordinate 0 = item A0
    const uint64_t A0 = c->pb_carry_n ? c->pb_carry_abs0 : c->pb_total;
    const uint64_t S1 = c->pb_total + n, nn = S1 - A0;
    c->h_packets.clear();
    c->h_tags.clear();
    c->h_bursts.clear();
    if (nn) {
        ENSURE(c, c->bb, (nn + pad) * sizeof(float));
        ENSURE(c, c->avg, (nn + pad) * sizeof(float));
        float *bb = (float *)c->bb.p, *avg = (float *)c->avg.p;
        if (c->pb_carry_n) {
            HIPCHK(c, hipMemcpyAsync(bb, c->pb_bb.p, c->pb_carry_n * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipMemcpyAsync(avg, c->pb_avg.p, c->pb_carry_n * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
        }
        if (n) {
            HIPCHK(c, hipMemcpyAsync(bb + c->pb_carry_n, in, n * sizeof(float), kind, c->stream));
            HIPCHK(c, hipMemcpyAsync(avg + c->pb_carry_n, inavg, n * sizeof(float), kind, c->stream));
        }
        ZERO_TAIL(c, 0, bb, nn, pad);
        ZERO_TAIL(c, 1, avg, nn, pad);
        // positions that can be decided now (as am_process_iq: a hit decided now must also be one if the stream ended here)
        const uint64_t P0 = c->pb_next;
        uint64_t P1 = P0, emit_max_abs = ~(uint64_t)0;
        if (flush) {
            uint64_t em;
            if (flush_limits(c, S1, &em)) {
                emit_max_abs = em;
                if (em + 1 > P0) P1 = em + 1;
            }
        } else {
            const uint64_t hold = (uint64_t)(AM_BURST + 4) * (uint64_t)c->spc_hi;
            if (S1 > hold && S1 - hold > P0) P1 = S1 - hold;
        }
        if (P1 > P0) {
            uint32_t M = 0, fin = 0;
            int rc = run_candidates(c, bb, avg, (uint32_t)(P0 - A0), (uint32_t)(P1 - A0), &M);
            if (rc != AM_OK) return rc;
            const uint32_t cur0 = c->pb_cur > A0 ? (uint32_t)std::min<uint64_t>(c->pb_cur - A0, 0xFFFFFFF0u) : 0u;
            const uint32_t emax = emit_max_abs == ~(uint64_t)0 ? 0xFFFFFFFFu : (uint32_t)(emit_max_abs - A0);
            fin = cur0;
            rc = run_chain_and_slice(c, bb, avg, M, cur0, emax, A0, true, &fin,
                                     (uint32_t)((P1 - P0) / ((uint64_t)AM_BURST * (uint64_t)c->spc) + 2));
            if (rc != AM_OK) return rc;
            if (A0 + fin > c->pb_cur) c->pb_cur = A0 + fin;
            c->pb_next = P1;
        }
        if (!flush) {
            // what the next call's decisions still read: everything from the first undecided position on
            const uint64_t C0 = c->pb_next > A0 ? c->pb_next : A0;
            const uint64_t keep = S1 - C0;
            if (keep) {
                ENSURE(c, c->pb_bb, keep * sizeof(float));
                ENSURE(c, c->pb_avg, keep * sizeof(float));
                HIPCHK(c, hipMemcpyAsync(c->pb_bb.p, bb + (C0 - A0), keep * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
                HIPCHK(c, hipMemcpyAsync(c->pb_avg.p, avg + (C0 - A0), keep * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
            }
            c->pb_carry_abs0 = C0;
            c->pb_carry_n = keep;
            c->pb_total = S1;
        }
        HIPCHK(c, hipStreamSynchronize(c->stream));          // (the caller may reuse its buffers)
    }
    const uint64_t nt = c->h_tags.size();
    if (flush) {
        // (the hits were copied to the host vectors above; the stream starts over at item 0)
        std::vector<am_tag> keep_t = c->h_tags;
        std::vector<float> keep_b = c->h_bursts;
        reset_stream(c);
        c->h_tags.swap(keep_t);
        c->h_bursts.swap(keep_b);
    }
    if (n_out) *n_out = nt;
    if (nt > cap) return fail(c, AM_ECAPACITY, "burst/tag arrays too small");
    if (nt) {
        if (!bursts || !tags) return fail(c, AM_EINVAL, "null output");
        memcpy(bursts, c->h_bursts.data(), nt * AM_BURST * sizeof(float));
        memcpy(tags, c->h_tags.data(), nt * sizeof(am_tag));
    }
    return AM_OK;
}

int am_slicer_work(am_ctx *c, const float *bursts, const am_tag *tags, uint64_t nb, uint32_t flags,
                   am_packet *out, uint64_t cap, uint64_t *n_out)
{
    if (!c) return AM_EINVAL;
    if (n_out) *n_out = 0;
    if (nb && (!bursts || !tags)) return fail(c, AM_EINVAL, "null input");
    if (nb == 0) return AM_OK;
    if (nb > 0x7FFFFFFFu) return fail(c, AM_EINVAL, "too many bursts");
    HIPCHK(c, hipSetDevice(c->device));
    const hipMemcpyKind kind = (flags & AM_F_DEVICE_IN) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    ENSURE(c, c->bursts, nb * AM_BURST * sizeof(float));
    ENSURE(c, c->tags, nb * sizeof(am_tag));
    ENSURE(c, c->packets, nb * sizeof(am_packet));
    HIPCHK(c, hipMemcpyAsync(c->bursts.p, bursts, nb * AM_BURST * sizeof(float), kind, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->tags.p, tags, nb * sizeof(am_tag), kind, c->stream));
    if (int rc = ensure_scalars(c); rc != AM_OK) return rc;
    const uint32_t nb32 = (uint32_t)nb;
    HIPCHK(c, hipMemcpyAsync(c->scalars.p, &nb32, sizeof(nb32), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, am_launch_slice((float *)c->bursts.p, (am_tag *)c->tags.p, (const uint32_t *)c->scalars.p, nb32,
                              (uint32_t *)c->crc_pow.p, (am_packet *)c->packets.p, nullptr, nullptr, c->stream));
    c->h_packets.resize(nb);
    HIPCHK(c, hipMemcpyAsync(c->h_packets.data(), c->packets.p, nb * sizeof(am_packet), hipMemcpyDeviceToHost,
                             c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->pending.clear();
    collect_accepted(c);
    return hand_out(c, out, cap, n_out);
}

// modes_crc.cc:55-63 semantics (generator 0xFFF409, zero start value), byte-serial
uint32_t am_crc24(const uint8_t *data, int nbytes)
{
    struct Table {
        uint32_t v[256];
        Table()
        {
            for (uint32_t b = 0; b < 256; b++) {
                uint32_t r = b << 16;
                for (int k = 0; k < 8; k++) r = (r & 0x800000u) ? ((r << 1) ^ 0xFFF409u) : (r << 1);
                v[b] = r & 0xFFFFFFu;
            }
        }
    };
    static const Table tab;                                     // (function-local static: initialised once, thread-safe)
    const uint32_t *table = tab.v;
    uint32_t r = 0;
    for (int i = 0; i < nbytes; i++) r = ((r << 8) ^ table[((r >> 16) ^ data[i]) & 0xFFu]) & 0xFFFFFFu;
    return r;
}

int am_format_message(const am_packet *p, int first, char *buf, size_t cap)
{
    if (!p || !buf) return AM_EINVAL;
    char tmp[192];
    int w = 0;
    const int nb = p->nbytes <= 14 ? p->nbytes : 14;
    for (int m = 0; m < nb; m++) w += snprintf(tmp + w, sizeof(tmp) - (size_t)w, "%02x", (unsigned)p->data[m]);
    // slicer_impl.cc:191-192: setw(6)/setfill('0') hex crc, then the reference level with the
    // stream's current precision (6 before the first setprecision(10), 10 ever after)
    w += snprintf(tmp + w, sizeof(tmp) - (size_t)w, " %06x %.*g %llu %.10g", (unsigned)p->crc, first ? 6 : 10,
                  (double)p->ref, (unsigned long long)p->secs, p->frac);
    if ((size_t)w + 1 > cap) return AM_ECAPACITY;
    memcpy(buf, tmp, (size_t)w + 1);
    return w;
}

// A batch of messages per call: text k starts at buf + offsets[k] (NUL-terminated), offsets[n] = bytes used.  `first` applies
// to packet 0 only (the member ostringstream's precision is 6 until the first message has been formatted, 10 ever after:
// slicer_impl.cc:186-194, slicer_impl.h:43).  Nothing is written beyond cap; AM_ECAPACITY leaves the bytes needed in *need.
int am_format_messages(const am_packet *pkts, uint64_t n, int first, char *buf, size_t cap, uint64_t *offsets, uint64_t *need)
{
    if (need) *need = 0;
    if (n && (!pkts || !offsets)) return AM_EINVAL;
    if (!buf && cap) return AM_EINVAL;
    uint64_t used = 0;
    bool fits = true;
    char tmp[200];
    for (uint64_t k = 0; k < n; ++k) {
        const int w = am_format_message(pkts + k, (first && k == 0) ? 1 : 0, tmp, sizeof(tmp));
        if (w < 0) return w;
        if (fits && used + (uint64_t)w + 1 <= (uint64_t)cap) {
            offsets[k] = used;
            memcpy(buf + used, tmp, (size_t)w + 1);
        } else
            fits = false;
        used += (uint64_t)w + 1;
    }
    if (need) *need = used;
    if (!fits) return AM_ECAPACITY;
    if (offsets) offsets[n] = used;
    return AM_OK;
}

int am_shard_halo(const am_ctx *c, uint64_t *left, uint64_t *right)
{
    if (!c || !left || !right) return AM_EINVAL;
    const uint64_t S = (uint64_t)c->spc;
    *left = 2 * (uint64_t)AM_CHIPS_AVG * S + S;     // up to one block (alignment) + one block + one chip
    if (c->use_dcblock) *left += am_dcblock_history(c->spc);
    *right = (uint64_t)(AM_BURST + 4) * (uint64_t)c->spc_hi;   // late shift + 240-chip burst
    return AM_OK;
}

// am_shard_scan, or (msg_dev != null) its host-free form: everything is enqueued, the exit table goes to the DEVICE
// message msg_dev = {count, -} + msg_cap entries, nothing is waited for (a first step without a candidate-density
// estimate still reads the count back once).
static int shard_scan_core(am_ctx *c, const float *iq, uint64_t abs_start, uint64_t abs_end, uint64_t total_n,
                           uint32_t flags, am_shard_exit *table, uint64_t cap, uint64_t *n_table, am_shard_exit *msg_dev,
                           uint64_t msg_cap)
{
    if (!c) return AM_EINVAL;
    if (n_table) *n_table = 0;
    if (abs_end < abs_start || abs_end > total_n) return fail(c, AM_EINVAL, "bad chunk bounds");
    HIPCHK(c, hipSetDevice(c->device));
    c->shard_ready = false;
    uint64_t hl, hr;
    am_shard_halo(c, &hl, &hr);
    const uint64_t S = (uint64_t)c->spc, L = (uint64_t)AM_CHIPS_AVG * S, LH = L + S;
    const uint64_t src_abs0 = abs_start > hl ? abs_start - hl : 0;
    const uint64_t src_abs1 = std::min(total_n, abs_end + hr);
    const uint64_t nsrc = src_abs1 - src_abs0;
    if (nsrc > ((uint64_t)1 << 31)) return fail(c, AM_EINVAL, "chunk larger than 2^31 samples");
    if (nsrc && !iq) return fail(c, AM_EINVAL, "null iq");
    HIPCHK(c, hipEventRecord(c->ev[0], c->stream));
    c->total_pending = false;
    c->dom_timed = false;
    const float *src = iq;
    if (!(flags & AM_F_DEVICE_IN) && nsrc) {
        ENSURE(c, c->src, nsrc * 2 * sizeof(float));
        HIPCHK(c, hipMemcpyAsync(c->src.p, iq, nsrc * 2 * sizeof(float), hipMemcpyHostToDevice, c->stream));
        src = (const float *)c->src.p;
    }
    // positions this chunk owns; the global end-of-stream rule bounds the last chunk -- unless the stream goes on (AM_F_MORE:
    // total_n = samples so far; the chunk must come with its whole right halo)
    uint64_t em = 0;
    uint64_t P0 = abs_start, P1 = abs_start;
    c->shard_more = (flags & AM_F_MORE) != 0;
    if (c->shard_more) {
        if (abs_end + hr > total_n) return fail(c, AM_EINVAL, "AM_F_MORE: the chunk needs its whole right halo");
        P1 = abs_end;
    } else if (flush_limits(c, total_n, &em)) P1 = std::max(P0, std::min(abs_end, em + 1));
    const uint64_t out_abs0 = (abs_start / L) * L;
    const uint64_t need0 = out_abs0 > LH ? out_abs0 - LH : 0;
    uint64_t fsrc_abs0 = src_abs0;
    {
        int rc = apply_dcblock(c, &src, &fsrc_abs0, src_abs1, need0);
        if (rc != AM_OK) return rc;
    }
    if (fsrc_abs0 > need0) return fail(c, AM_EINVAL, "internal: left halo too short");
    const uint64_t out_n = src_abs1 - out_abs0;
    const uint64_t pad = zero_pad(c->spc_hi);
    uint32_t M = 0;
    c->spec_now = false;
    c->Mdev = nullptr;
    if (P1 > P0 && out_n) {
        const bool generic = scan_path(c, false) == 1;
        ENSURE(c, c->bb, (out_n + pad) * sizeof(float));
        float *bb = (float *)c->bb.p, *avg = nullptr;
        ZERO_TAIL(c, 0, bb, out_n, pad);
        if (generic) {
            ENSURE(c, c->avg, (out_n + pad) * sizeof(float));
            avg = (float *)c->avg.p;
            ZERO_TAIL(c, 1, avg, out_n, pad);
        }
        int rc = run_front_and_candidates(c, src, fsrc_abs0, src_abs1, out_abs0, out_n, bb, avg,
                                          (uint32_t)(P0 - out_abs0), (uint32_t)(P1 - out_abs0), &M, true);
        if (rc != AM_OK) return rc;
    }
    c->shard_base = out_abs0;
    c->shard_start = abs_start;
    c->shard_end = abs_end;
    c->shard_total = total_n;
    // exit table for the candidates the scan can enter at: those in the first 241*spc samples of
    // the chunk (the farthest a predecessor's skip can reach) and the first one after them
    const uint64_t lead = (uint64_t)c->geom.B + (uint64_t)c->geom.late_max + 1;   // (241 spc + 1 for whole samples per chip)
    const uint64_t lead_end = abs_start + lead;                 // absolute, exclusive
    uint32_t n_dev = 0;
    if (msg_dev) {
        // host-free: the table (what fits the message) and its count go to the device message; whether the capacity
        // this scan was launched for sufficed comes back with the resolve step's completion ticket
        const uint32_t *Mp = c->spec_now ? c->Mdev : nullptr;
        int rc = chain_prepare(c, M, true, Mp);
        if (rc != AM_OK) return rc;
        n_dev = (uint32_t)std::min<uint64_t>(std::min<uint64_t>(M, lead + 1), msg_cap);
        if (int rce = ensure_shard_exit(c); rce != AM_OK) return rce;
        // the message header: {entries, overflow}, {where the scan left this chunk a step ago, -} -- written by the table kernel,
        // or (no candidate at all) by a one-wave launch
        if (!n_dev) HIPCHK(c, am_launch_shard_header(msg_dev, (const uint64_t *)c->shard_exit.p, c->stream));
        if (n_dev)
            HIPCHK(c, am_launch_chain_exit_table((uint32_t *)c->pos.p, (uint32_t *)c->tgt.p, M, n_dev,
                                                 (uint32_t)std::min<uint64_t>(lead_end - out_abs0, 0xFFFFFFFFull),
                                                 (uint32_t *)c->cscratch.p, out_abs0, msg_dev + AM_SHARD_MSG_HEADER, c->stream, Mp,
                                                 msg_dev, (const uint64_t *)c->shard_exit.p));
        c->last_M = M;
        c->shard_ready = true;
        return AM_OK;
    }
    for (int attempt = 0; attempt < 2; ++attempt) {
        const uint32_t *Mp = c->spec_now ? c->Mdev : nullptr;
        int rc = chain_prepare(c, M, true, Mp);
        if (rc != AM_OK) return rc;
        n_dev = (uint32_t)std::min<uint64_t>(M, lead + 1);
        uint32_t actual = M;
        if (!c->pin_scalars) {
            HIPCHK(c, hipHostMalloc((void **)&c->pin_scalars, 16 * sizeof(uint32_t), hipHostMallocCoherent | hipHostMallocMapped));
            memset(c->pin_scalars, 0, 16 * sizeof(uint32_t));
        }
        if (n_dev) {
            // the table goes straight to pinned host memory; the host then reads it up to its last entry
            if (c->pin_exit_cap < n_dev) {
                if (c->pin_exit) (void)hipHostFree(c->pin_exit);
                c->pin_exit = nullptr; c->pin_exit_cap = 0;
                HIPCHK(c, hipHostMalloc((void **)&c->pin_exit, ((size_t)n_dev + 64) * sizeof(am_shard_exit),
                                        hipHostMallocCoherent | hipHostMallocMapped));
                c->pin_exit_cap = n_dev + 64;
            }
            HIPCHK(c, am_launch_chain_exit_table((uint32_t *)c->pos.p, (uint32_t *)c->tgt.p, M, n_dev,
                                                 (uint32_t)std::min<uint64_t>(lead_end - out_abs0, 0xFFFFFFFFull),
                                                 (uint32_t *)c->cscratch.p, out_abs0, c->pin_exit, c->stream, Mp));
        }
        // one completion ticket (with the device-side candidate count of a capacity launch) instead of
        // copies through the runtime and a stream synchronisation
        c->pin_scalars[3] = M;
        const uint32_t seq = ++c->ticket_seq;
        HIPCHK(c, am_launch_ticket(c->pin_scalars + 8, seq, c->stream, Mp, c->pin_scalars + 3));
        HIPCHK(c, hipEventRecord(c->ev[2], c->stream));
        c->total_pending = true;
        HIPCHK(c, wait_for_ticket(c, seq));
        actual = c->pin_scalars[3];
        if (!Mp || actual <= M) { c->last_M = actual; break; }
        // more candidates than the capacity this scan was launched for: once more with the exact count
#if defined(AM_TEST_KNOBS)
        if (getenv("AIRMODES_TRACE_SPEC")) fprintf(stderr, "airmodes: shard capacity %u < %u candidates, scan redone\n", M, actual);
#endif
        rc = run_refine(c, c->ref_bb, c->ref_avg, c->ref_nseg, c->ref_stride, c->ref_mode, &M, c->ref_endj, 0);
        if (rc != AM_OK) return rc;
    }
    c->spec_density = (P1 > P0) ? (double)c->last_M / (double)(P1 - P0) : 0.0;
    c->last_dom_ms = 0.0f;
    if (c->dom_timed && hipEventElapsedTime(&c->last_dom_ms, c->ev[3], c->ev[1]) != hipSuccess) c->ht[6] += 1.0;
    uint64_t nt = 0;
    // (a capacity launch may have written an end marker {pos = ~0} behind the last real candidate: not an entry)
    const uint32_t n_real = (uint32_t)std::min<uint64_t>(n_dev, c->last_M);
    for (uint32_t i = 0; i < n_real; i++) {
        if (c->pin_exit[i].pos == ~(uint64_t)0) break;
        nt = i + 1;
        if (c->pin_exit[i].pos >= lead_end) break;              // first candidate past the lead-in: last entry
    }
    c->shard_ready = true;
    if (n_table) *n_table = nt;
    if (nt > cap) return fail(c, AM_ECAPACITY, "exit table too small");
    if (nt) {
        if (!table) return fail(c, AM_EINVAL, "null table");
        memcpy(table, c->pin_exit, (size_t)nt * sizeof(am_shard_exit));
    }
    return AM_OK;
}

int am_shard_scan(am_ctx *c, const float *iq, uint64_t abs_start, uint64_t abs_end, uint64_t total_n,
                  uint32_t flags, am_shard_exit *table, uint64_t cap, uint64_t *n_table)
{
    return shard_scan_core(c, iq, abs_start, abs_end, total_n, flags, table, cap, n_table, nullptr, 0);
}

// Is p memory the device can address (device, managed, or pinned / registered host memory)?  Pageable host memory handed to
// the host-free calls would be a GPU memory fault, not an error code.
static bool device_addressable(const void *p)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged || a.type == hipMemoryTypeHost;
}

int am_shard_scan_async(am_ctx *c, const float *iq, uint64_t abs_start, uint64_t abs_end, uint64_t total_n,
                        uint32_t flags, am_shard_exit *msg_dev, uint64_t msg_cap)
{
    if (!c || !msg_dev || msg_cap == 0) return AM_EINVAL;
    if (!device_addressable(msg_dev)) return fail(c, AM_EINVAL, "am_shard_scan_async: msg_dev is not device-addressable memory");
    return shard_scan_core(c, iq, abs_start, abs_end, total_n, flags, nullptr, 0, nullptr, msg_dev, msg_cap);
}

int am_shard_entry(const am_shard_exit *const *tables, const uint64_t *counts, const uint64_t *starts,
                   uint32_t nranks, uint64_t *entry)
{
    if (!entry || (nranks && (!tables || !counts || !starts))) return AM_EINVAL;
    uint64_t cur = 0;                                           // the scan starts at sample 0
    for (uint32_t r = 0; r < nranks; r++) {
        entry[r] = cur;
        const am_shard_exit *t = tables[r];
        const uint64_t n = counts[r];
        // first candidate of chunk r at or after cur; the table covers every position cur can take
        uint64_t i = 0;
        while (i < n && t[i].pos < cur) i++;
        if (i < n) cur = t[i].exit > cur ? t[i].exit : cur;     // no candidate left: the scan passes through
    }
    (void)starts;
    return AM_OK;
}

int am_shard_entry2(const am_shard_exit *const *tables, const uint64_t *counts, uint32_t nranks, uint64_t cur_in,
                    uint64_t *entry, uint64_t *leave)
{
    if (nranks && (!tables || !counts || (!entry && !leave))) return AM_EINVAL;
    uint64_t cur = cur_in;                                      // where the scan left the last chunk of the step before
    for (uint32_t r = 0; r < nranks; r++) {
        if (entry) entry[r] = cur;
        const am_shard_exit *t = tables[r];
        const uint64_t n = counts[r];
        uint64_t i = 0;
        while (i < n && t[i].pos < cur) i++;
        if (i < n) cur = t[i].exit > cur ? t[i].exit : cur;
        if (leave) leave[r] = cur;
    }
    return AM_OK;
}

int am_shard_get_exit(am_ctx *c, uint64_t *pos)
{
    if (!c || !pos) return AM_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    if (int rce = ensure_shard_exit(c); rce != AM_OK) return rce;
    HIPCHK(c, hipMemcpyAsync(pos, c->shard_exit.p, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return AM_OK;
}

int am_shard_set_exit(am_ctx *c, uint64_t pos)
{
    if (!c) return AM_EINVAL;
    HIPCHK(c, hipSetDevice(c->device));
    if (int rce = ensure_shard_exit(c); rce != AM_OK) return rce;
    HIPCHK(c, hipMemcpyAsync(c->shard_exit.p, &pos, sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return AM_OK;
}

int am_shard_keep_tail(am_ctx *c, void *dst, const void *src, uint64_t nbytes)
{
    if (!c || (nbytes && (!dst || !src))) return AM_EINVAL;
    c->keep_dst = dst; c->keep_src = src; c->keep_bytes = nbytes;
    return AM_OK;
}

int am_stream_copy(am_ctx *c, void *dst, const void *src, uint64_t nbytes)
{
    if (!c || (nbytes && (!dst || !src))) return AM_EINVAL;
    if (!nbytes) return AM_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(dst, src, nbytes, hipMemcpyDeviceToDevice, c->stream));
    return AM_OK;
}

int am_shard_resolve(am_ctx *c, uint64_t cur_in, am_packet *out, uint64_t cap, uint64_t *n_out)
{
    if (!c) return AM_EINVAL;
    if (n_out) *n_out = 0;
    if (!c->shard_ready) return fail(c, AM_EINVAL, "am_shard_scan has not been called");
    HIPCHK(c, hipSetDevice(c->device));
    c->pending.clear();
    c->last_tags = 0;
    uint64_t em = 0;
    if (c->chain_M == 0 || (!c->shard_more && (!flush_limits(c, c->shard_total, &em) || em < c->shard_base))) {
        if (c->keep_bytes) {                                    // (nothing to slice: the tail is still kept)
            HIPCHK(c, hipMemcpyAsync(c->keep_dst, c->keep_src, c->keep_bytes, hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
        return AM_OK;
    }
    const uint32_t cur0 = cur_in > c->shard_base ? (uint32_t)std::min<uint64_t>(cur_in - c->shard_base, 0xFFFFFFF0u) : 0u;
    const uint32_t emax = c->shard_more ? 0xFFFFFFFEu : (uint32_t)std::min<uint64_t>(em - c->shard_base, 0xFFFFFFFEu);
    uint32_t fin = 0;
    const uint32_t max_hits = (uint32_t)((c->shard_end - c->shard_start + (uint64_t)c->spc) /
                                         ((uint64_t)AM_BURST * (uint64_t)c->spc) + 2);
    c->resolving_shard = true;
    int rc = chain_finish(c, (const float *)c->bb.p, cur0, emax, c->shard_base, false, &fin, max_hits);
    c->resolving_shard = false;
    if (rc != AM_OK) return rc;
    c->last_tags = c->n_hits;
    return hand_out(c, out, cap, n_out);
}

int am_shard_resolve_async(am_ctx *c, const am_shard_exit *msgs_dev, uint32_t world, uint32_t rank, uint64_t msg_cap,
                           am_packet *out, uint64_t cap, uint64_t *n_out, int *redo)
{
    if (!c || !redo || (world && !msgs_dev) || rank >= world) return AM_EINVAL;
    if (n_out) *n_out = 0;
    *redo = 0;
    if (!c->shard_ready) return fail(c, AM_EINVAL, "am_shard_scan_async has not been called");
    if (!device_addressable(msgs_dev)) return fail(c, AM_EINVAL, "am_shard_resolve_async: msgs_dev is not device-addressable memory");
    HIPCHK(c, hipSetDevice(c->device));
    c->pending.clear();
    c->last_tags = 0;
    uint64_t em = 0;
    if (int rc = ensure_scalars(c); rc != AM_OK) return rc;
    uint32_t *cur0_dev = (uint32_t *)c->scalars.p + 4, *flag_dev = (uint32_t *)c->scalars.p + 5;
    // the entry position of this chunk is composed from everybody's exit tables on the device: by the block walk itself, or
    // (nothing to slice here) by a launch of its own -- the other ranks must still learn whether a table overflowed: it did
    // so on every rank alike
    // (the flag is written, 0 or 1, by whichever kernel composes the entry: no fill in front of it)
    if (int rce = ensure_shard_exit(c); rce != AM_OK) return rce;
    if (c->chain_M == 0 || (!c->shard_more && (!flush_limits(c, c->shard_total, &em) || em < c->shard_base))) {
        HIPCHK(c, am_launch_shard_entry(msgs_dev, world, rank, (uint32_t)msg_cap, c->shard_base, cur0_dev, flag_dev,
                                        (uint64_t *)c->shard_exit.p, c->stream));
        uint32_t f = 0;
        if (c->keep_bytes) HIPCHK(c, hipMemcpyAsync(c->keep_dst, c->keep_src, c->keep_bytes, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(&f, flag_dev, sizeof(f), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        *redo = f ? 1 : 0;
        return AM_OK;
    }
    const uint32_t emax = c->shard_more ? 0xFFFFFFFEu : (uint32_t)std::min<uint64_t>(em - c->shard_base, 0xFFFFFFFEu);
    uint32_t fin = 0;
    const uint32_t max_hits = (uint32_t)((c->shard_end - c->shard_start + (uint64_t)c->spc) /
                                         ((uint64_t)AM_BURST * (uint64_t)c->spc) + 2);
    am_entry_src es;
    es.msgs = msgs_dev; es.world = world; es.rank = rank; es.cap = (uint32_t)msg_cap; es.base_abs = c->shard_base; es.flags = flag_dev;
    es.exit_out = (uint64_t *)c->shard_exit.p;
    es.cur_in = nullptr;                                        // (the scan position comes from the last rank's header)
    c->entry_src = &es;
    c->flag_src = flag_dev;
    c->resolving_shard = true;
    int rc = chain_finish(c, (const float *)c->bb.p, 0, emax, c->shard_base, false, &fin, max_hits);
    c->resolving_shard = false;
    c->entry_src = nullptr;
    c->flag_src = nullptr;
    // the dominant kernel's event pair of this step's scan (am_shard_scan_async only enqueued): both events lie in front of
    // the completion ticket chain_finish waited for
    c->last_dom_ms = 0.0f;
    if ((rc == AM_OK || rc == AM_RETRY_EXACT) && c->dom_timed &&
        hipEventElapsedTime(&c->last_dom_ms, c->ev[3], c->ev[1]) != hipSuccess) c->ht[6] += 1.0;
    if (rc == AM_RETRY_EXACT) { c->pending.clear(); *redo = 1; return AM_OK; }   // more candidates than the capacity the scan was launched for
    if (rc != AM_OK) return rc;
    if (c->pin_scalars[4]) { c->pending.clear(); *redo = 1; return AM_OK; }       // a table did not fit its message
    c->spec_density = (c->shard_end > c->shard_start) ? (double)c->last_M / (double)(c->shard_end - c->shard_start) : 0.0;
    c->last_tags = c->n_hits;
    return hand_out(c, out, cap, n_out);
}

} // extern "C"

/* ---- ONE continuing stream, several of its chunks in flight ---------------------------------------------------
 * The reference block is a streaming block: general_work() resumes where the last call stopped, for ever (lib/preamble_impl.cc:
 * 139-246, consume_each at :213,237,244).  am_process_iq does that one chunk at a time -- the host's turn-around and the launch-bound
 * tail of chunk k in front of the streaming kernel of chunk k + 1.  Here consecutive chunks of ONE stream are in flight on one GPU,
 * each on a context and a stream of its own, with the time-shard machinery at world 1: chunk k decides the positions
 * [S_k - H, S_k + n_k - H) from its own samples and the tail of the chunk before it (AM_F_MORE); its scan -- front end, refinement,
 * block exits, exit table -- does not depend on where the greedy scan enters the chunk and is enqueued at once; its resolve step
 * takes the entry position ON THE DEVICE from the word in which the chunk before it leaves it (am_entry_src::cur_in), ordered
 * behind that chunk's resolve by an event.  Nothing waits for the host between two chunks; item counts and time stamps keep counting
 * (positions are stream-absolute).  A chunk whose exit table did not fit its message, or whose scan met more candidates than the
 * capacity it was launched for, is flagged in its message header: the chunks behind it (which composed their entry from a word that
 * was not written) are drained, the chunk is redone on the synchronous path (host tables, am_shard_entry2), and the drained chunks
 * are submitted again.  Packets of all chunks, concatenated, == am_process_iq over the same cuts == the oracle over the stream. */
struct am_spipe_slot {
    am_ctx *c = nullptr;
    am_shard_exit *msg = nullptr;       // device message of the chunk's scan: header + msg_cap entries
    hipEvent_t done = nullptr;          // behind the chunk's resolve step
    const float *iq = nullptr;          // the chunk as submitted
    uint64_t S = 0, n = 0;
    bool flush = false, first = false;
};
struct am_spipe {
    std::vector<am_spipe_slot> slot;
    size_t head = 0, inflight = 0;
    uint64_t S_next = 0;                // samples of the stream submitted so far
    bool ended = false;                 // a flushing chunk is in flight: the stream starts over once everything is collected
    uint64_t hl = 0, H = 0;             // history in front of a position / look-ahead behind it
    uint32_t msg_cap = 512;
    uint64_t *zero_dev = nullptr;       // a device word that holds 0: where the scan "left the chunk before" the stream's first
    uint64_t exit_host = 0;             // where the scan left the last collected chunk (host copy: a redone chunk starts there)
    const float *prev_iq = nullptr;     // the chunk submitted last
    uint64_t prev_n = 0;
    int prev_slot = -1;
    uint64_t redone = 0;                // chunks that went through the synchronous path
    std::vector<am_shard_exit> host_tab;
    const am_ctx *last_fail = nullptr;
    char err[160] = "";
};

static int spipe_fail(am_spipe *p, int code, const char *what)
{
    p->last_fail = nullptr;
    snprintf(p->err, sizeof(p->err), "%s", what);
    return code;
}

// what the time-shard calls need of chunk (S, n) of the stream: the positions it decides and the samples it hands over
static void spipe_bounds(const am_spipe *p, const am_spipe_slot &sl, uint64_t *a0, uint64_t *a1, uint64_t *total, const float **ptr)
{
    *total = sl.S + sl.n;
    *a0 = sl.first ? 0 : sl.S - p->H;
    *a1 = sl.flush ? *total : sl.S + sl.n - p->H;
    const uint64_t lo = *a0 > p->hl ? *a0 - p->hl : 0;          // first sample the library wants
    *ptr = sl.iq - (sl.S - lo) * 2;
}

// The resolve step of the resident chunk, ENQUEUED only (am_spipe, am_shard_resolve_submit): entry position composed on the device
// from `world` messages -- starting from *cur_in where given --, marking, extraction, slicing, one completion ticket; c->pend says
// what shard_resolve_complete has to wait for.  walk_done: recorded behind the block walk (or behind the entry kernel of a chunk
// with nothing to slice): what a later chunk's resolve step has to wait for.
static int shard_resolve_enqueue(am_ctx *c, const am_shard_exit *msgs, uint32_t world, uint32_t rank, uint32_t msg_cap,
                                 const uint64_t *cur_in, uint64_t *carry_out, hipEvent_t walk_done)
{
    c->pending.clear();
    c->last_tags = 0;
    if (int rcs = ensure_scalars(c); rcs != AM_OK) return rcs;
    if (int rce = ensure_shard_exit(c); rce != AM_OK) return rce;
    uint32_t *cur0_dev = (uint32_t *)c->scalars.p + 4, *flag_dev = (uint32_t *)c->scalars.p + 5;
    if (!c->pin_scalars) {
        HIPCHK(c, hipHostMalloc((void **)&c->pin_scalars, 16 * sizeof(uint32_t), hipHostMallocCoherent | hipHostMallocMapped));
        memset(c->pin_scalars, 0, 16 * sizeof(uint32_t));
    }
    am_ctx::Pending &P = c->pend;
    P = am_ctx::Pending();
    uint64_t em = 0;
    if (c->chain_M == 0 || (!c->shard_more && (!flush_limits(c, c->shard_total, &em) || em < c->shard_base))) {
        // nothing to slice: the entry is still composed (the chunk passes the scan position on), one ticket
        HIPCHK(c, am_launch_shard_entry(msgs, world, rank, msg_cap, c->shard_base, cur0_dev, flag_dev, (uint64_t *)c->shard_exit.p, c->stream, cur_in,
                                        carry_out));
        if (walk_done) HIPCHK(c, hipEventRecord(walk_done, c->stream));
        if (c->keep_bytes) HIPCHK(c, hipMemcpyAsync(c->keep_dst, c->keep_src, c->keep_bytes, hipMemcpyDeviceToDevice, c->stream));
        const uint32_t seq = ++c->ticket_seq;
        HIPCHK(c, am_launch_ticket(c->pin_scalars + 8, seq, c->stream, flag_dev, c->pin_scalars + 4, (const uint64_t *)c->shard_exit.p,
                                   reinterpret_cast<uint64_t *>(c->pin_scalars + 12)));
        P.active = true; P.scanned = false; P.seq = seq;
        return AM_OK;
    }
    const uint32_t emax = c->shard_more ? 0xFFFFFFFEu : (uint32_t)std::min<uint64_t>(em - c->shard_base, 0xFFFFFFFEu);
    uint32_t fin = 0;
    const uint32_t max_hits = (uint32_t)((c->shard_end - c->shard_start + (uint64_t)c->spc) / ((uint64_t)AM_BURST * (uint64_t)c->spc) + 2);
    am_entry_src es;
    es.msgs = msgs; es.world = world; es.rank = rank; es.cap = msg_cap; es.base_abs = c->shard_base; es.flags = flag_dev;
    es.exit_out = (uint64_t *)c->shard_exit.p; es.cur_in = cur_in; es.carry_out = carry_out;
    c->entry_src = &es;
    c->flag_src = flag_dev;
    c->word_src = (const uint64_t *)c->shard_exit.p;
    c->walk_event = walk_done;
    c->resolving_shard = true;
    c->defer = true;
    int rc = chain_finish(c, (const float *)c->bb.p, 0, emax, c->shard_base, false, &fin, max_hits);
    c->defer = false;
    c->resolving_shard = false;
    c->entry_src = nullptr;
    c->flag_src = nullptr;
    c->word_src = nullptr;
    c->walk_event = nullptr;
    if (rc != AM_DEFERRED) return rc == AM_OK ? fail(c, AM_EHIP, "internal: the resolve step was not deferred") : rc;
    P.active = true;                                            // (chain_finish filled in scanned / seq / M / Mp / n_max)
    return AM_OK;
}

// ... and its completion: waits for the ticket; the packets stay in the context's hand-out list.  *redo: a table did not fit its message
// or a scan outgrew its capacity (every rank reads the same headers: every rank gets the same answer).  *exit_after: where the scan
// left this chunk
static int shard_resolve_complete(am_ctx *c, int *redo, uint64_t *exit_after)
{
    am_ctx::Pending &P = c->pend;
    *redo = 0;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, wait_for_ticket(c, P.seq));
    int rc = AM_OK;
    if (P.scanned) {
        uint32_t fin = 0;
        rc = chain_collect(c, P.M, P.Mp, P.n_max, false, &fin);
        if (rc == AM_RETRY_EXACT) { *redo = 1; rc = AM_OK; }
    } else
        c->tail_synced = true;
    P.active = false; P.scanned = false;
    if (rc != AM_OK) return rc;
    if (c->pin_scalars[4]) *redo = 1;                           // (the header said so)
    if (*redo) c->pending.clear();
    if (exit_after) *exit_after = *reinterpret_cast<volatile uint64_t *>(c->pin_scalars + 12);
    c->spec_density = (c->shard_end > c->shard_start) ? (double)c->last_M / (double)(c->shard_end - c->shard_start) : 0.0;
    c->last_tags = c->n_hits;
    c->last_dom_ms = 0.0f;
    if (c->dom_timed && hipEventElapsedTime(&c->last_dom_ms, c->ev[3], c->ev[1]) != hipSuccess) c->ht[6] += 1.0;
    return AM_OK;
}

// enqueue scan + resolve of the chunk in slot k (its fields are filled in); nothing waits
static int spipe_enqueue(am_spipe *p, size_t k)
{
    am_spipe_slot &sl = p->slot[k];
    am_ctx *c = sl.c;
    HIPCHK(c, hipSetDevice(c->device));
    uint64_t a0, a1, total;
    const float *ptr;
    spipe_bounds(p, sl, &a0, &a1, &total, &ptr);
    // the samples in front of the chunk: the tail of the chunk before it, copied there unless the stream is contiguous in memory
    if (!sl.first && p->prev_iq) {
        const uint64_t front = p->hl + p->H;
        const float *src = p->prev_iq + (p->prev_n - front) * 2;
        float *dst = const_cast<float *>(sl.iq) - front * 2;
        if (src != dst) HIPCHK(c, hipMemcpyAsync(dst, src, front * 2 * sizeof(float), hipMemcpyDeviceToDevice, c->stream));
    }
    int rc = shard_scan_core(c, ptr, a0, a1, total, AM_F_DEVICE_IN | (sl.flush ? 0u : AM_F_MORE), nullptr, 0, nullptr, sl.msg, p->msg_cap);
    if (rc != AM_OK) return rc;
    // the resolve step: behind the resolve of the chunk before (the word it leaves), entry composed on the device
    const uint64_t *cur_in = p->zero_dev;
    if (!sl.first && p->prev_slot >= 0) {
        am_spipe_slot &pv = p->slot[(size_t)p->prev_slot];
        if (int rce = ensure_shard_exit(pv.c); rce != AM_OK) return rce;
        cur_in = (const uint64_t *)pv.c->shard_exit.p;
        HIPCHK(c, hipStreamWaitEvent(c->stream, pv.done, 0));
    }
    return shard_resolve_enqueue(c, sl.msg, 1, 0, p->msg_cap, cur_in, nullptr, sl.done);
}

// wait for the chunk in slot k; its packets stay in the context's hand-out list.  *redo: the chunk must go through the synchronous path
static int spipe_complete(am_spipe *p, size_t k, int *redo, uint64_t *exit_after)
{
    return shard_resolve_complete(p->slot[k].c, redo, exit_after);
}

extern "C" {

am_spipe *am_spipe_create(int device, double rate, float threshold_db, int use_pmf, int use_dcblock, int depth, int *err)
{
    if (depth < 1 || depth > 16) { if (err) *err = AM_EINVAL; return nullptr; }
    am_spipe *p = new (std::nothrow) am_spipe();
    if (!p) { if (err) *err = AM_ENOMEM; return nullptr; }
    p->slot.resize((size_t)depth);
    for (int k = 0; k < depth; k++) {
        am_spipe_slot &sl = p->slot[(size_t)k];
        sl.c = am_create(device, rate, threshold_db, use_pmf, use_dcblock, err);
        if (!sl.c) { am_spipe_destroy(p); return nullptr; }
        // (am_pipe left one of a CU's six persistent front-end slots free for the other batches' small kernels until round 5; with
        // am_k_refine_seg behind the front end -- 36 KB of LDS per workgroup -- a fifth of the LDS buys nothing: six, measured,
        // profiles/r6_tail/ab_5_pipes_fe3_wgs_per_cu.txt)
        if (hipMalloc((void **)&sl.msg, ((size_t)AM_SHARD_MSG_HEADER + p->msg_cap) * sizeof(am_shard_exit)) != hipSuccess ||
            hipEventCreateWithFlags(&sl.done, hipEventDisableTiming) != hipSuccess) {
            if (err) *err = AM_ENOMEM;
            am_spipe_destroy(p);
            return nullptr;
        }
    }
    uint64_t hl = 0, hr = 0;
    am_shard_halo(p->slot[0].c, &hl, &hr);
    p->hl = hl; p->H = hr;
    if (hipMalloc((void **)&p->zero_dev, 2 * sizeof(uint64_t)) != hipSuccess || hipMemset(p->zero_dev, 0, 2 * sizeof(uint64_t)) != hipSuccess) {
        if (err) *err = AM_ENOMEM;
        am_spipe_destroy(p);
        return nullptr;
    }
    p->host_tab.resize((size_t)(241 * (p->slot[0].c->spc_hi) + 8));
    if (err) *err = AM_OK;
    return p;
}

void am_spipe_destroy(am_spipe *p)
{
    if (!p) return;
    for (am_spipe_slot &sl : p->slot) {
        if (sl.c) { (void)hipSetDevice(sl.c->device); (void)hipStreamSynchronize(sl.c->stream); }
        if (sl.msg) (void)hipFree(sl.msg);
        if (sl.done) (void)hipEventDestroy(sl.done);
        am_destroy(sl.c);
    }
    if (p->zero_dev) (void)hipFree(p->zero_dev);
    delete p;
}

int am_spipe_depth(const am_spipe *p) { return p ? (int)p->slot.size() : AM_EINVAL; }
int am_spipe_in_flight(const am_spipe *p) { return p ? (int)p->inflight : AM_EINVAL; }
uint64_t am_spipe_redone(const am_spipe *p) { return p ? p->redone : 0; }

int am_spipe_front(const am_spipe *p, uint64_t *front)
{
    if (!p || !front) return AM_EINVAL;
    *front = p->hl + p->H;
    return AM_OK;
}

int am_spipe_set_rx_time(am_spipe *p, uint64_t offset, uint64_t secs, double frac)
{
    if (!p) return AM_EINVAL;
    if (p->inflight) return spipe_fail(p, AM_EINVAL, "rx_time: collect the chunks in flight first (the tag tables are read by their kernels)");
    for (am_spipe_slot &sl : p->slot)
        if (int rc = am_set_rx_time(sl.c, offset, secs, frac); rc != AM_OK) { p->last_fail = sl.c; return rc; }
    return AM_OK;
}

int am_spipe_submit(am_spipe *p, const float *iq, uint64_t n, uint32_t flags)
{
    if (!p) return AM_EINVAL;
    if (p->inflight == p->slot.size()) return spipe_fail(p, AM_ECAPACITY, "every context of the pipe has a chunk in flight: collect the oldest one first");
    if (p->ended) return spipe_fail(p, AM_EINVAL, "the stream was flushed: collect its chunks before the next stream starts");
    if (!iq || n == 0) return spipe_fail(p, AM_EINVAL, "null or empty chunk");
    const bool first = p->S_next == 0;
    const uint64_t front = p->hl + p->H;
    if (n < front + 1 || n > ((uint64_t)1 << 30)) return spipe_fail(p, AM_EINVAL, "a chunk must hold more samples than am_spipe_front() reports (and fewer than 2^30)");
    const size_t k = (p->head + p->inflight) % p->slot.size();
    am_spipe_slot &sl = p->slot[k];
    sl.iq = iq; sl.S = p->S_next; sl.n = n; sl.flush = (flags & AM_F_FLUSH) != 0; sl.first = first;
    const int rc = spipe_enqueue(p, k);
    if (rc != AM_OK) { p->last_fail = sl.c; return rc; }
    p->inflight++;
    p->S_next += n;
    p->prev_iq = iq; p->prev_n = n; p->prev_slot = (int)k;
    if (sl.flush) p->ended = true;
    return AM_OK;
}

int am_spipe_collect(am_spipe *p, am_packet *out, uint64_t cap, uint64_t *n_out)
{
    if (!p) return AM_EINVAL;
    if (n_out) *n_out = 0;
    if (p->inflight == 0) return spipe_fail(p, AM_EINVAL, "no chunk in flight");
    const size_t k = p->head;
    am_spipe_slot &sl = p->slot[k];
    am_ctx *c = sl.c;
    if (c->pend.active) {
        int redo = 0;
        uint64_t leave = 0;
        int rc = spipe_complete(p, k, &redo, &leave);
        if (rc != AM_OK) { p->last_fail = c; return rc; }
        if (redo) {
            // the chunks behind this one composed their entry from a word that was never written: drain them, redo this chunk with
            // the tables on the host, submit them again
            p->redone++;
            const size_t D = p->slot.size();
            for (size_t j = 1; j < p->inflight; j++) {
                int r2 = 0;
                uint64_t l2 = 0;
                am_ctx *cj = p->slot[(k + j) % D].c;
                rc = spipe_complete(p, (k + j) % D, &r2, &l2);
                cj->pending.clear();
                if (rc != AM_OK) { p->last_fail = cj; return rc; }
            }
            uint64_t a0, a1, total, m = 0;
            const float *ptr;
            spipe_bounds(p, sl, &a0, &a1, &total, &ptr);
            rc = am_shard_scan(c, ptr, a0, a1, total, AM_F_DEVICE_IN | (sl.flush ? 0u : AM_F_MORE), p->host_tab.data(), p->host_tab.size(), &m);
            if (rc != AM_OK) { p->last_fail = c; return rc; }
            const am_shard_exit *tabs[1] = {p->host_tab.data()};
            uint64_t entry = 0;
            rc = am_shard_entry2(tabs, &m, 1, p->exit_host, &entry, &leave);
            if (rc != AM_OK) return spipe_fail(p, rc, "am_shard_entry2");
            uint64_t got = 0;
            rc = am_shard_resolve(c, entry, nullptr, 0, &got);   // (the packets stay in the context's hand-out list: AM_ECAPACITY is expected)
            if (rc != AM_OK && rc != AM_ECAPACITY) { p->last_fail = c; return rc; }
            rc = am_shard_set_exit(c, leave);
            if (rc != AM_OK) { p->last_fail = c; return rc; }
            // the chunks behind it, in order, once more (chunk k's own word now holds where the scan left it)
            const float *piq = sl.iq;
            uint64_t pn = sl.n;
            int pslot = (int)k;
            for (size_t j = 1; j < p->inflight; j++) {
                const size_t kj = (k + j) % D;
                p->prev_iq = piq; p->prev_n = pn; p->prev_slot = pslot;
                rc = spipe_enqueue(p, kj);
                if (rc != AM_OK) { p->last_fail = p->slot[kj].c; return rc; }
                piq = p->slot[kj].iq; pn = p->slot[kj].n; pslot = (int)kj;
            }
            p->prev_iq = piq; p->prev_n = pn; p->prev_slot = pslot;
        }
        p->exit_host = leave;
    }
    const int hrc = hand_out(c, out, cap, n_out);
    if (hrc == AM_ECAPACITY) { p->last_fail = c; return hrc; }  // the packets stay: call again with a larger array
    p->head = (p->head + 1) % p->slot.size();
    p->inflight--;
    if (p->ended && p->inflight == 0) {
        // the stream is over: the next submit starts a new one at sample 0
        for (am_spipe_slot &s2 : p->slot) { am_reset(s2.c); (void)am_shard_set_exit(s2.c, 0); }
        p->S_next = 0; p->ended = false; p->prev_iq = nullptr; p->prev_n = 0; p->prev_slot = -1; p->exit_host = 0;
    }
    return hrc;
}

int am_shard_resolve_submit(am_ctx *c, const am_shard_exit *msgs_dev, uint32_t world, uint32_t rank, uint64_t msg_cap,
                            const uint64_t *cur_in_dev, uint64_t *carry_out_dev)
{
    if (!c || (world && !msgs_dev) || rank >= world) return AM_EINVAL;
    if (!c->shard_ready) return fail(c, AM_EINVAL, "am_shard_scan_async has not been called");
    if (c->pend.active) return fail(c, AM_EINVAL, "a submitted resolve step has not been collected (am_shard_resolve_collect)");
    if (!device_addressable(msgs_dev) || (cur_in_dev && !device_addressable(cur_in_dev)) || (carry_out_dev && !device_addressable(carry_out_dev)))
        return fail(c, AM_EINVAL, "am_shard_resolve_submit: a pointer is not device-addressable memory");
    HIPCHK(c, hipSetDevice(c->device));
    return shard_resolve_enqueue(c, msgs_dev, world, rank, (uint32_t)msg_cap, cur_in_dev, carry_out_dev, nullptr);
}

int am_shard_resolve_collect(am_ctx *c, am_packet *out, uint64_t cap, uint64_t *n_out, int *redo)
{
    if (!c || !redo) return AM_EINVAL;
    if (n_out) *n_out = 0;
    if (!c->pend.active) {                                      // (an AM_ECAPACITY left the packets behind: hand them out now)
        *redo = 0;
        return hand_out(c, out, cap, n_out);
    }
    if (int rc = shard_resolve_complete(c, redo, nullptr); rc != AM_OK) return rc;
    if (*redo) return AM_OK;
    return hand_out(c, out, cap, n_out);
}

const char *am_spipe_last_error(const am_spipe *p)
{
    if (!p || p->slot.empty()) return g_create_err;
    return p->last_fail ? p->last_fail->err : p->err;
}

float am_spipe_last_kernel_ms(const am_spipe *p)
{
    if (!p || p->slot.empty()) return 0.0f;
    const size_t last = (p->head + p->slot.size() - 1) % p->slot.size();
    return p->slot[last].c->last_dom_ms;
}

} // extern "C"

extern "C" {
/* ---- several batches in flight from one host thread ------------------------------------------------------- */
struct am_pipe {
    std::vector<am_ctx *> sub;
    size_t head = 0, inflight = 0;
    const am_ctx *last_fail = nullptr;   // the sub-context whose call failed last (am_pipe_last_error)
    char err[128] = "";                  // ... or the pipe's own message
};

am_pipe *am_pipe_create(int device, double rate, float threshold_db, int use_pmf, int use_dcblock, int depth, int *err)
{
    if (depth < 1 || depth > 16) { if (err) *err = AM_EINVAL; return nullptr; }
    am_pipe *p = new (std::nothrow) am_pipe();
    if (!p) { if (err) *err = AM_ENOMEM; return nullptr; }
    for (int k = 0; k < depth; k++) {
        am_ctx *c = am_create(device, rate, threshold_db, use_pmf, use_dcblock, err);
        if (!c) { am_pipe_destroy(p); return nullptr; }
        // batches in flight: the streaming kernel of one batch leaves room on every CU (LDS, registers) for the small kernels
        // of the others -- five persistent workgroups per CU instead of six (measured: 285-294 -> 297-302 GS/s at depth 4)
        // (round 6: all six persistent front-end workgroups per CU here as well -- see am_spipe_create)
#if defined(AM_TEST_KNOBS) && !defined(AM_HIP_EMULATION)
        // measurement only (test builds): every context of the pipe on its own share of the CUs (hipExtStreamCreateWithCUMask),
        // so that batches overlap on the CUs instead of in the gaps -- VERDICT r4 #3; result in profiles/r5_fe64
        if (const char *e = getenv("AIRMODES_PIPE_CU_PARTS")) {
            const int parts = atoi(e), cus = am_device_cus();
            if (parts > 1 && parts <= depth && cus >= parts) {
                uint32_t mask[32] = {0};
                const int lo = (k % parts) * cus / parts, hi = ((k % parts) + 1) * cus / parts;
                for (int b = lo; b < hi && b < 1024; ++b) mask[b >> 5] |= 1u << (b & 31);
                hipStream_t st = nullptr;
                if (hipExtStreamCreateWithCUMask(&st, (uint32_t)((cus + 31) / 32), mask) == hipSuccess) {
                    (void)hipStreamDestroy(c->own_stream);
                    c->own_stream = st;
                    c->stream = st;
                    c->fe_wgs_per_cu = 0;
                }
            }
        }
#endif
        p->sub.push_back(c);
    }
    if (err) *err = AM_OK;
    return p;
}

void am_pipe_destroy(am_pipe *p)
{
    if (!p) return;
    for (am_ctx *c : p->sub) am_destroy(c);
    delete p;
}

int am_pipe_depth(const am_pipe *p) { return p ? (int)p->sub.size() : AM_EINVAL; }
int am_pipe_in_flight(const am_pipe *p) { return p ? (int)p->inflight : AM_EINVAL; }

int am_pipe_submit(am_pipe *p, const float *iq, uint64_t n, uint32_t flags)
{
    if (!p) return AM_EINVAL;
    if (p->inflight == p->sub.size()) {
        p->last_fail = nullptr;
        snprintf(p->err, sizeof(p->err), "every context of the pipe has a batch in flight: collect the oldest one first");
        return AM_ECAPACITY;
    }
    am_ctx *c = p->sub[(p->head + p->inflight) % p->sub.size()];
    const int rc = am_submit_iq(c, iq, n, flags | AM_F_FLUSH);
    if (rc == AM_OK) p->inflight++;
    else p->last_fail = c;
    return rc;
}

// K whole streams in ONE scan of the pipe's next free context (am_submit_multi); collected like any other batch, the packets stream by
// stream; am_pipe_multi_counts then says how many each stream of the scan collected LAST got.
int am_pipe_submit_multi(am_pipe *p, float *iq, uint32_t k, const uint64_t *n, uint32_t flags)
{
    if (!p) return AM_EINVAL;
    if (p->inflight == p->sub.size()) {
        p->last_fail = nullptr;
        snprintf(p->err, sizeof(p->err), "every context of the pipe has a batch in flight: collect the oldest one first");
        return AM_ECAPACITY;
    }
    am_ctx *c = p->sub[(p->head + p->inflight) % p->sub.size()];
    const int rc = am_submit_multi(c, iq, k, n, flags);
    if (rc == AM_OK) p->inflight++;
    else p->last_fail = c;
    return rc;
}

int am_pipe_multi_counts(am_pipe *p, uint64_t *count, uint32_t k)
{
    if (!p || p->sub.empty()) return AM_EINVAL;
    am_ctx *c = p->sub[(p->head + p->sub.size() - 1) % p->sub.size()];   // the batch collected last
    const int rc = am_multi_counts(c, count, k);
    if (rc != AM_OK) p->last_fail = c;
    return rc;
}

int am_pipe_collect(am_pipe *p, am_packet *out, uint64_t cap, uint64_t *n_out)
{
    if (!p) return AM_EINVAL;
    if (n_out) *n_out = 0;
    if (p->inflight == 0) {
        p->last_fail = nullptr;
        snprintf(p->err, sizeof(p->err), "no batch in flight");
        return AM_EINVAL;
    }
    am_ctx *c = p->sub[p->head];
    const int rc = am_collect(c, out, cap, n_out);
    if (rc != AM_OK) p->last_fail = c;
    if (rc == AM_ECAPACITY) return rc;                              // the packets stay: call again with a larger array
    p->head = (p->head + 1) % p->sub.size();
    p->inflight--;
    return rc;
}

const char *am_pipe_last_error(const am_pipe *p)
{
    if (!p || p->sub.empty()) return g_create_err;
    return p->last_fail ? p->last_fail->err : p->err;              // the message of whatever failed last
}

float am_pipe_last_kernel_ms(const am_pipe *p)
{
    if (!p || p->sub.empty()) return 0.0f;
    return p->sub[(p->head + p->sub.size() - 1) % p->sub.size()]->last_dom_ms;   // the batch collected last
}

const char *am_last_error(const am_ctx *c) { return c ? c->err : g_create_err; }

int am_last_timing(am_ctx *c, float *total_ms, float *dom_ms)
{
    if (!c) return AM_EINVAL;
    if (total_ms) {
        if (c->total_pending) {
            // the end-of-work event of the last scan was queued behind its completion ticket
            (void)hipSetDevice(c->device);
            if (hipEventSynchronize(c->ev[2]) == hipSuccess)
                (void)hipEventElapsedTime(&c->last_total_ms, c->ev[0], c->ev[2]);
            c->total_pending = false;
        }
        *total_ms = c->last_total_ms;
    }
    if (dom_ms) *dom_ms = c->last_dom_ms;
    return AM_OK;
}

int am_last_frontend(const am_ctx *c) { return c ? c->last_fe : AM_EINVAL; }

long long am_last_num_candidates(const am_ctx *c)
{
    return c ? (long long)c->last_M : (long long)AM_EINVAL;
}

} // extern "C"
