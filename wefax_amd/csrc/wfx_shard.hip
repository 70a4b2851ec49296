// One capture decoded by several GPUs: the exact path of wefax.py: Demodulator.process() sharded by sample range.
// (include/wefax_hip.h, section "one capture over several GPUs"; the distributed transforms are in wfx_dist.hip, the
// communicator in wfx_comm.hip.)
//
// Rank r owns the rows [rows[r], rows[r+1]) of the [R1][M] arrangement of the L = N/2 packed points of the Hilbert
// transform, i.e. the samples [2 rows[r] M, 2 rows[r+1] M) at 11 025 Hz, and -- with a resampler in front -- the same rows
// of the input's packed points.  Stages and what they exchange (all on the context's stream, no host round trip):
//
//   [a4 merge]  [a5 resample: distributed rfft -> bin copy -> distributed irfft, halo of 32 samples delivered with it]
//   a6 notch on own samples + 32 (49-tap form; filtfilt's exact edges on the ranks that hold a true end)
//   a7 distributed Hilbert convolution (4 transposes), |x + iH| + median 5 on the block (halo of 2 samples)
//   a8 percentiles: level-0 histogram (fused) -> ALL-REDUCE -> level 1 -> ALL-REDUCE -> candidates -> ALL-GATHER -> finish
//      (every rank ends with the same low / high on its device); quantise own block
//   ONE GATHER of the uint8 stream (1 byte per sample) to rank 0
//   a9 sync search + a10 bicubic image on rank 0 (17 + 60 us of work for a 60-minute capture: not worth a second exchange)
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "wfx_dist.h"

//
// Round 4, the COLUMNS layout (shard_plan::cols): the stencil stages run where the transforms' first / last passes
// leave their data -- rank r holds, of every row n1 of the [R1][M] arrangement, the columns [cols[r], cols[r+1]): R1 segments of
// samples, each with a halo from the neighbouring ranks' columns (24 + 2 samples of audio for the notch, 2 points of the Hilbert
// transform for the median) -- so the rows <-> columns transposes E1 and E4 of every transform disappear: 4 array transposes per
// decode instead of 8 (2 instead of 4 without a resampler), plus two halo exchanges of a few KB.
//   [a4 merge] [a5: first pass on the caller's columns, E2, slab passes, bin copy, slab passes, E3, last pass, HALO exchange]
//   a6 notch on the R1 segments (one launch over the rows laid end to end; true edges on rank 0's first / the last rank's last)
//   a7 first pass in place, E2, slab passes x spectrum, E3, last pass, HALO exchange, |x + iH| + median 5 per segment
//   a8 as before; the gather delivers every rank's segments to rank 0, which puts them in order (one 2-D copy) -- a9, a10
#define SH_HALO 32          // samples of audio kept beyond the own range on either side (>= 24 notch taps + 2 median)
#define SH_VHALO 2          // points of the Hilbert transform delivered beyond the own rows
#define SH_HALO_END 192     // columns layout of a PADDED form: the capture ends inside a segment, and filtfilt's exact edge there
                            // is made of the last 127 samples -- the ranks whose own samples lie within 64 of the end must see it
#define SH_FMM_HR 320       // plan 3: raw samples handed over beyond the own range on either side, round the circle (a leaf of <= 64 for the near field + the
                            // notch's 24 taps; at the capture's ends the exact filtfilt edge is made of 127 + 9 samples and the segment form wants >= 254)
#define SH_FMM_HF (SH_FMM_HR - 24)      // filtered samples kept beyond the own range
#define SH_RS_HS 64         // plan 3 in front of a resampler: input samples beyond the own sources on either side (one leaf of the resampler tree)
#define SH_CAND_CAP 4096    // least number of candidate keys per query and rank that travel in the all-gather

struct shard_plan {
    int world = 1, rank = 0;
    wfx_dist_geom g;
    // padded forms over more than one rank: `g` deals only the rows that hold samples (wfx_dist_geom::rows_used) -- every rank owns an
    // equal share of the capture, E1 / E4 carry half the bytes; the kernel's own transform (first decode) has taps in ALL rows and
    // runs on `gk`, the full-rows geometry, through a transform object of its own that is released afterwards
    wfx_dist_geom gk;
    bool split_kernel = false;
    bool resample = false;
    int in_kind = 0;
    uint64_t n0 = 0, n = 0;
    long long M1 = 0, K = 0, M1s = 0, Ms = 0;
    // any even length (round 3): when K = n / 2 has no distributed plan the Hilbert convolution is embedded in a transform of
    // Kp >= 2K - 1 points (13-smooth, radix pairs only); the rows of THAT arrangement that hold samples are dealt to the ranks in
    // equal shares (`g.rows_used`; with WFX_SHARD_ALL_ROWS, or one rank, all rows are: the ranks whose rows are all padding then own
    // nothing and still take part in every transform and collective)
    bool padded = false;
    // odd lengths (no resampling): nothing to pack -- one point per sample (K = n), real rows (mr2_pass IN_MODE 3), a real kernel,
    // always the padded form (Kp >= 2n - 1: twice the points and four times the exchanged bytes of an even capture of that length)
    bool plain = false;
    int spp = 2;                          // samples per point
    // No distributed form (a RESAMPLED capture whose half-lengths are odd or not 13-smooth multiples of a common first radix --
    // its inverse transform's length is the reference's int(11025 n0 / fs), not ours to choose -- or a capture too short for
    // the world size): rank 0 owns the whole capture and decodes it alone with the fused one-GPU path, the other ranks own
    // nothing and only receive the scalars.  Not a stopgap: the distributed form of such a capture would be two Bluestein
    // convolutions (2.5-3x the points of the packed transforms, 8 more exchanges), and the cost model of DESIGN 6.6 puts that
    // behind one GPU at every world size up to 8.
    bool single = false;
    char single_reason[160] = {0};
    long long Kp = 0;
    int wrap_rank = 0;                    // the rank holding the capture's last pair: it needs V[0] from rank 0 as "V[K]"
    uint64_t own_lo = 0, own_hi = 0, in_lo = 0, in_hi = 0;
    uint64_t seg_lo = 0, seg_hi = 0;      // samples at 11 025 Hz held for the notch
    // columns layout (round 4): this rank's columns of the K-point arrangement (Hilbert transform, resampler inverse) and of the
    // M1-point one (resampler forward); hs = samples of halo on either side of a segment of audio (0 with one rank: the rows are
    // dense and the neighbours in memory ARE the halo)
    bool cols = false;
    long long cH0 = 0, wH = 0, cF0 = 0, wF = 0;
    int hs = 0;
    long long xrs = 0;                    // samples between the starts of two rows of audio: 2 wH + 2 hs
    // k1 subsets per rank: E2, the slab passes and E3 of every distributed transform run subset by subset, the exchange of one (on
    // the communicator's own stream) overlapping the passes of another.  4 where the first radix has enough outputs for
    // world * 4 subsets, 1 with one rank; WFX_SHARD_CHUNKS overrides (every rank must see the same value)
    int nchunk = 1;
    // plan 3 (round 6): chunk-local fast multipole Hilbert transform (wfx_fmm.hip) -- every rank runs the leaf-level kernels on its own
    // boxes of the gather level `fg.lg` (hence a contiguous range of leaf workgroups, hence of samples); what travels is the gather level's
    // weights, three boxes per finer level and side, four envelope values per seam -- then the select's collectives and the one gather
    bool fmm = false;
    wfx_fmm_shard_geo fg{};
    long long gb_lo = 0, gb_hi = 0;
    // ... in front of it, for a capture at another rate, the resampler's multipole form on ITS tree (sources: the n0 input samples): the ranks
    // are dealt boxes of a level `lgc` at or above both trees' gather levels, so that a rank's arc of the circle is the same in both --
    // its resampled samples are the Hilbert tree's own samples, and only the 320 beyond either end travel
    bool rs = false;
    wfx_fmm_shard_geo fgr{};
    long long gbr_lo = 0, gbr_hi = 0;
    int lgc = 0;
    int in_halo = 0;                      // input frames held beyond [in_lo, in_hi) on either side, round the circle
    // the cost model's verdict
    int forced = 0;
    double model_single = 0, model_comp = 0, model_wire = 0;
    unsigned long long model_bytes = 0;
};

// ---- cost model of a sharded decode on one node -------------------------------------------------------------
// One GPU (measured, MI355X): the exact path takes ~46 ps per 11 025 Hz sample while its arrays fit the 256 MiB Infinity Cache
// (configs[1]: 0.33 ms), ~70 ps beyond (configs[2] / [3]), plus ~8.7 ps per input frame when a resampler runs in front.
// N ranks: that work / N, one copy of the rank's share per surviving packing step, and the exchanges: in a transpose over N ranks
// a rank sends array / N^2 to each peer, every peer over its own xGMI link, so an exchange takes (array / N^2) / B_link + latency.
// B_link defaults to 50 GB/s (what grouped send / recv reaches of a link's 153 GB/s peak); WFX_LINK_GBS overrides it.
static double link_gbs()
{
    const char *e = getenv("WFX_LINK_GBS");
    const double v = e ? atof(e) : 0.0;
    return v > 0.0 ? v : 50.0;
}

// per-exchange launch latency the model charges (a grouped send / recv of RCCL: 10-30 us); WFX_LINK_LAT_US overrides
static double link_lat_s()
{
    const char *e = getenv("WFX_LINK_LAT_US");
    const double us = e ? atof(e) : 20.0;
    return (us > 0.0 && us < 1e4 ? us : 20.0) * 1e-6;
}

static void plan_cost(shard_plan &pl, bool cols)
{
    const int W = pl.world;
    const double n = (double)pl.n, n0 = (double)pl.n0;
    const double per = n * 16.0 <= 256e6 ? 46e-12 : 70e-12;
    // (a length whose half is not 13-smooth runs its convolution on twice the points: 1.55x measured on one GPU, odd lengths 1.63x)
    pl.model_single = 20e-6 + per * n * (pl.plain ? 1.63 : (pl.padded ? 1.55 : 1.0)) + (pl.resample ? 8.7e-12 * n0 : 0.0);
    const double K16 = 16.0 * (double)pl.Kp, M16 = 16.0 * (double)pl.M1;
    const double in_es = pl.in_kind == WFX_IN_I16_MONO ? 4.0 : 16.0;
    // array transposes: rows layout E1..E4 per transform (a padded form's E1 / E4 carry the sample-bearing rows only), columns
    // layout E2 / E3 only
    const double used = (!cols && pl.g.rows_used > 0 && pl.g.R1 > 0) ? (double)pl.g.rows_used / (double)pl.g.R1 : 1.0;
    double arrays = cols ? 2.0 * K16 : (2.0 + 2.0 * used) * K16;
    int nex = cols ? 2 : 4, ncopy = cols ? 1 : 3;
    if (pl.resample) {
        arrays += cols ? (M16 + K16) : (M16 * (in_es / 16.0) + M16 + 2.0 * K16);
        nex += cols ? 2 : 4;
        ncopy += cols ? 1 : 3;
    }
    const double lat = link_lat_s(), bl = link_gbs() * 1e9;
    const double small = (cols ? (pl.resample ? 2 : 1) : 0) + 3;          // halo exchanges, two all-reduces, one all-gather
    // columns layout: every transpose travels as C k1 subsets on the communicator's own stream while the slab passes of the subsets
    // already there run (about 45 % of a rank's transform work sits in slab passes): (C - 1) / C of the shorter of the two is hidden
    // -- and every subset is an exchange of its own, with its own launch latency
    const int C = cols ? std::max(1, pl.nchunk) : 1;
    const double t_arrays = W > 1 ? arrays / ((double)W * W) / bl : 0.0;
    const double hidden = C > 1 ? (double)(C - 1) / C * std::min(0.45 * pl.model_single / W, t_arrays) : 0.0;
    pl.model_wire = W > 1 ? t_arrays - hidden + (nex * C + small) * lat + (n / W) / bl + lat : 0.0;
    pl.model_comp = pl.model_single / W + ncopy * (32.0 * (double)pl.Kp / W) / 4e12 + (W > 1 ? 80e-6 : 0.0);      // + rank 0's tail: sync search, image
    pl.model_bytes = W > 1 ? (unsigned long long)(arrays * (W - 1) / W + n * (W - 1) / W) : 0ull;
}

static int make_plan(wfx_ctx *ctx, const wfx_decode_params *p, int world, int rank, shard_plan &pl)
{
    if (!p) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null parameters");
    if (world < 1 || rank < 0 || rank >= world) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "bad rank %d for world size %d", rank, world);
    if (p->in_kind != WFX_IN_I16_MONO && p->in_kind != WFX_IN_F64_MONO && p->in_kind != WFX_IN_I16_STEREO)
        return wfx_fail(ctx, WFX_ERR_BAD_ARG, "sharded decode: input kind %d", p->in_kind);
    if (p->hilbert_mode != WFX_HILBERT_FFT) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "sharded decode: the exact Hilbert mode only");
    if (p->n == 0 || p->n0 == 0 || (!p->resample && p->n != p->n0)) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "bad capture lengths");
    if (p->n > (1ull << 31) || p->n0 >= (1ull << 32)) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "capture too long");
    if (p->width <= 0) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "bad line width");
    pl.world = world;
    pl.rank = rank;
    pl.resample = p->resample != 0;
    pl.in_kind = p->in_kind;
    pl.n0 = p->n0;
    pl.n = p->n;
    auto single = [&](const char *why) {
        pl.single = true;
        pl.padded = pl.plain = false;
        snprintf(pl.single_reason, sizeof pl.single_reason, "%s", why);
        pl.g = wfx_dist_geom();
        pl.g.world = world;
        pl.g.rank = rank;
        pl.own_lo = rank == 0 ? 0 : pl.n;
        pl.own_hi = pl.n;
        pl.in_lo = rank == 0 ? 0 : pl.n0;
        pl.in_hi = pl.n0;
        pl.seg_lo = pl.own_lo;
        pl.seg_hi = pl.own_hi;
        return 0;
    };
    const int want = p->shard_plan & 15;               // 0 auto, 1 distributed, 2 single, 3 chunk-local multipole form
    const bool want_rows = (p->shard_plan & 16) != 0 || getenv("WFX_SHARD_ROWS") != nullptr;
    pl.forced = want != 0;
    if (want == 2) return single("asked for by the caller (wfx_decode_params.shard_plan)");
    if (want == 3) {
        // ---- plan 3: the capture cut into contiguous ranges of leaf workgroups of the multipole tree (geometry: a function of n alone) ----
        if (p->resample) {
            if (wfx_rs_shard_geometry(p->n0, p->n, &pl.fgr) != 0)
                return wfx_fail(ctx, WFX_ERR_BAD_ARG, "sharded decode, multipole plan: no multipole form of the resampler for %llu -> %llu samples (downsampling to an even count)",
                                (unsigned long long)p->n0, (unsigned long long)p->n);
            pl.rs = true;
        }
        if (wfx_fmm_shard_geometry(p->n, &pl.fg) != 0)
            return wfx_fail(ctx, WFX_ERR_BAD_ARG, "sharded decode, multipole plan: no multipole form for %llu samples (even, >= 32 768)", (unsigned long long)p->n);
        pl.lgc = pl.rs ? std::min(pl.fg.lg, pl.fgr.lg) : pl.fg.lg;
        const long long G = 1ll << pl.lgc;
        if (G < world) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "sharded decode, multipole plan: %lld boxes at the gather level for %d ranks", G, world);
        pl.fmm = true;
        pl.g = wfx_dist_geom();
        pl.g.world = world;
        pl.g.rank = rank;
        pl.gb_lo = ((long long)rank * G / world) << (pl.fg.lg - pl.lgc);
        pl.gb_hi = ((long long)(rank + 1) * G / world) << (pl.fg.lg - pl.lgc);
        const int sh = pl.fg.ltop - pl.fg.lg + 6;                          // gather box -> leaves
        pl.own_lo = (uint64_t)wfx_fmm_leaf_first_host(p->n, pl.fg.L, pl.gb_lo << sh);
        pl.own_hi = (uint64_t)wfx_fmm_leaf_first_host(p->n, pl.fg.L, pl.gb_hi << sh);
        pl.in_lo = pl.seg_lo = pl.own_lo;
        pl.in_hi = pl.seg_hi = pl.own_hi;
        pl.in_halo = SH_FMM_HR;
        if (pl.rs) {
            pl.gbr_lo = ((long long)rank * G / world) << (pl.fgr.lg - pl.lgc);
            pl.gbr_hi = ((long long)(rank + 1) * G / world) << (pl.fgr.lg - pl.lgc);
            const int shr = pl.fgr.ltop - pl.fgr.lg + 6;
            pl.in_lo = (uint64_t)wfx_fmm_leaf_first_host(p->n0, pl.fgr.L, pl.gbr_lo << shr);
            pl.in_hi = (uint64_t)wfx_fmm_leaf_first_host(p->n0, pl.fgr.L, pl.gbr_hi << shr);
            pl.in_halo = SH_RS_HS;
        }
        // model: the leaf-level kernels divide by the world size, the top is repeated by every rank; five small exchanges + the gather
        const double nn = (double)p->n;
        pl.model_single = 20e-6 + (nn * 16.0 <= 256e6 ? 46e-12 : 70e-12) * nn + (pl.rs ? 8.7e-12 * (double)p->n0 : 0.0);
        // (the resampler's multipole form: 52 ps per input sample on one GPU at the IQ hand-over's size, near field 2/3 of it)
        pl.model_comp = 60e-6 + (52e-12 * nn + (pl.rs ? 60e-6 * world + 52e-12 * (double)p->n0 : 0.0)) / world + (world > 1 ? 80e-6 : 0.0);
        const double lat = link_lat_s(), bl = link_gbs() * 1e9;
        pl.model_wire = world > 1 ? (pl.rs ? 8 : 6) * lat + (nn / world) / bl + lat : 0.0;
        pl.model_bytes = world > 1 ? (unsigned long long)(G * 256 * (world - 1) + (unsigned long long)world * (pl.fg.L - pl.fg.lg) * 6 * 256 + nn * (world - 1) / world) : 0ull;
        return 0;
    }
    if (p->n >= (1ull << 31)) return single("2^31 samples: beyond the distributed transforms' 32-bit indices");
    if (p->resample && ((p->n & 1) || (p->n0 & 1))) return single("a resampled capture with an odd sample count (its transforms are packed)");
    // odd length: real samples against scipy's real kernel on PACKED transforms (round 4; rounds 1-3: one complex point per sample,
    // twice the points) -- a point is a pair of samples here too, the last one half empty
    pl.plain = (p->n & 1) != 0;
    pl.spp = 2;
    pl.K = (long long)((p->n + 1) / 2);
    pl.M1 = (long long)(p->n0 / 2);
    long long lens[2] = {pl.K, pl.M1};
    int ra1 = 0, rb1 = 0;
    pl.Kp = pl.K;
    if (pl.plain || !wfx_dist_choose_r1(lens, pl.resample ? 2 : 1, world, &ra1, &rb1)) {
        // no plan for the capture's own half-length: pad the Hilbert convolution (a capture that is resampled on the way would
        // need the same for its two other transforms, whose lengths are the reference's to choose: not built)
        bool found = false;
        if (!pl.resample && pl.K >= 4096) {
            std::vector<long long> cand;
            wfx_mr_smooth_numbers(2 * pl.K - 1, (2 * pl.K - 1) + (2 * pl.K - 1) / 8, cand);
            // The padded length and its first radix are chosen the same way for every world size up to 8 (the plan of the largest:
            // what works for 8 ranks works for fewer), so that the float stages do not depend on the number of ranks here either;
            // the rank whose rows straddle the capture's end must be left with a workable number of samples (or none) -- first
            // for every world size up to 8, and only if no candidate manages that, for this one alone.
            const int wref = world > 8 ? world : 8;
            for (int strict = 1; strict >= 0 && !found; --strict)
                for (long long m : cand) {
                    if (m >= (1ll << 31)) break;
                    if (!wfx_dist_choose_r1(&m, 1, wref, &ra1, &rb1, true)) continue;
                    const int r1 = ra1 * rb1;
                    const long long ms = m / r1;
                    const long long rv = (pl.K + ms - 1) / ms;                    // rows that hold samples
                    bool fine = true;
                    for (int w = strict ? 1 : world; w <= (strict ? wref : world) && fine; ++w) {
                        if (w > 1 && rv < w) fine = false;
                        for (int r = 0; r < w && fine; ++r) {
                            const long long dealt = w > 1 ? rv : r1;                 // (one rank keeps all rows: its first pass reads them in place)
                            const long long a = (long long)r * dealt / w * ms, b = (long long)(r + 1) * dealt / w * ms;
                            const long long own = (b < pl.K ? b : pl.K) - (a < pl.K ? a : pl.K);
                            if (own < 2048) fine = false;
                        }
                    }
                    if (!fine) continue;
                    pl.Kp = m;
                    pl.padded = found = true;
                    break;
                }
        }
        if (!found)
            return single(pl.resample ? "a resampled capture whose half-lengths are not 13-smooth multiples of a common first radix"
                                      : "too short for a padded plan");
    }
    {
        const long long ms = pl.Kp / (ra1 * rb1);
        const int rv = (int)((pl.K + ms - 1) / ms);
        pl.split_kernel = pl.padded && world > 1 && rv < ra1 * rb1 && !WFX_LAB_ENV("WFX_SHARD_ALL_ROWS");      // (A/B switch: all R1 rows dealt, as before)
        if (!wfx_dist_make_geom(pl.g, world, rank, ra1, rb1, pl.split_kernel ? rv : 0)) return single("no geometry for this world size");
        if (pl.split_kernel && !wfx_dist_make_geom(pl.gk, world, rank, ra1, rb1)) return single("no geometry for this world size");
    }
    const int R1 = pl.g.R1;
    pl.Ms = pl.Kp / R1;
    pl.M1s = pl.M1 / R1;
    auto clipK = [&](long long pts) { return (uint64_t)(pts < pl.K ? pts : pl.K); };
    auto clipN = [&](uint64_t smp) { return smp < (uint64_t)p->n ? smp : (uint64_t)p->n; };        // (odd length: the last point holds one sample)
    pl.own_lo = clipN((uint64_t)pl.spp * clipK((long long)pl.g.rows[rank] * pl.Ms));
    pl.own_hi = clipN((uint64_t)pl.spp * clipK((long long)pl.g.rows[rank + 1] * pl.Ms));
    pl.wrap_rank = 0;
    for (int r = 0; r < world; ++r)
        if ((long long)pl.g.rows[r] * pl.Ms < pl.K) pl.wrap_rank = r;           // the last rank that owns samples
    pl.seg_lo = pl.own_lo >= SH_HALO ? pl.own_lo - SH_HALO : 0;
    pl.seg_hi = pl.own_hi + SH_HALO <= pl.n ? pl.own_hi + SH_HALO : pl.n;
    if (pl.own_hi == pl.own_lo) pl.seg_lo = pl.seg_hi = pl.own_lo;                // a rank of padding rows: nothing to filter
    // (the same verdict on every rank: the tests run over all of them)
    {
        // what wfx_dist::init asks of a transform of L points: every rank a row, and columns enough for the halo it delivers
        auto dist_ok = [&](long long L, int halo) {
            const long long M = L / R1;
            for (int d = 0; d < world; ++d) {
                const long long c0 = (long long)d * M / world / 4 * 4, c1 = d + 1 == world ? M : (long long)(d + 1) * M / world / 4 * 4;
                if (c1 - c0 < 2 || c1 - c0 < halo || pl.g.nrows(d) < 1) return false;
            }
            return true;
        };
        if (!dist_ok(pl.Kp, SH_VHALO) || (pl.resample && (!dist_ok(pl.M1, 0) || !dist_ok(pl.K, SH_HALO / 2)))) return single("too short for this world size");
    }
    for (int r = 0; r < world; ++r) {
        const uint64_t lo = clipN((uint64_t)pl.spp * clipK((long long)pl.g.rows[r] * pl.Ms)), hi = clipN((uint64_t)pl.spp * clipK((long long)pl.g.rows[r + 1] * pl.Ms));
        if ((r > 0 && hi > lo && lo < SH_HALO) || (hi - lo < 1024 && !(pl.padded && r > 0)) || (pl.padded && hi > lo && hi - lo < 64))
            return single("too short for this world size");
    }
    if (pl.resample) {
        pl.in_lo = 2ull * (uint64_t)pl.g.rows[rank] * (uint64_t)pl.M1s;
        pl.in_hi = 2ull * (uint64_t)pl.g.rows[rank + 1] * (uint64_t)pl.M1s;
    } else {
        pl.in_lo = pl.seg_lo;
        pl.in_hi = pl.seg_hi;
    }
    // ---- columns layout: unpadded even captures (every BASELINE size) and, since later in round 4, the any-length padded forms ----
    auto colrange = [&](long long M, int r, long long &c0, long long &w) {
        c0 = (long long)r * M / world / 4 * 4;
        const long long c1 = r + 1 == world ? M : (long long)(r + 1) * M / world / 4 * 4;
        w = c1 - c0;
    };
    // (padded forms -- arbitrary lengths at the native rate -- take it too when there is more than one rank; one rank keeps the rows
    // form, whose first pass reads the capture in place)
    const bool cols_padded = pl.padded && world > 1 && !pl.resample;
    bool cols_ok = !want_rows && ((!pl.padded && !pl.plain) || cols_padded);
    for (int r = 0; r < world && cols_ok; ++r) {
        long long c0, w;
        colrange(pl.Ms, r, c0, w);
        if (w < 128) cols_ok = false;                                          // a segment must hold the notch's edge block and its halo
        if (pl.resample) {
            colrange(pl.M1s, r, c0, w);
            if (w < 4) cols_ok = false;
        }
    }
    if (cols_ok) {
        pl.cols = true;
        if (pl.padded) {      // every rank holds its columns of ALL rows of the padded arrangement (the rows behind the capture are zeros)
            pl.split_kernel = false;
            if (!wfx_dist_make_geom(pl.g, world, rank, ra1, rb1, 0)) return single("no geometry for this world size");
        }
        colrange(pl.Ms, rank, pl.cH0, pl.wH);
        if (pl.resample) colrange(pl.M1s, rank, pl.cF0, pl.wF);
        pl.hs = world > 1 ? (pl.padded ? SH_HALO_END : SH_HALO) : 0;
        pl.xrs = 2 * pl.wH + 2 * pl.hs;
        pl.nchunk = 1;       // (chosen below, once the rest of the plan is known)
        pl.own_lo = 2ull * (uint64_t)pl.cH0;
        pl.own_hi = 2ull * (uint64_t)((long long)(R1 - 1) * pl.Ms + pl.cH0 + pl.wH);
        pl.in_lo = pl.resample ? 2ull * (uint64_t)pl.cF0 : pl.own_lo;
        pl.in_hi = pl.resample ? 2ull * (uint64_t)((long long)(R1 - 1) * pl.M1s + pl.cF0 + pl.wF) : pl.own_hi;
        pl.seg_lo = pl.own_lo;
        pl.seg_hi = pl.own_hi;
    }
    // ---- k1 subsets per rank: WFX_SHARD_CHUNKS, else what the model says is cheapest among 1 .. 4 (more subsets hide more slab
    // passes behind the wire, and pay one exchange latency each: long captures on few ranks take 4, short ones or many ranks 1 - 2) ----
    if (pl.cols) {
        auto feasible = [&](int c) {
            wfx_dist_geom gv;
            return c == 1 || wfx_dist_make_geom(gv, world * c, 0, pl.g.ra1, pl.g.rb1);
        };
        const char *e = getenv("WFX_SHARD_CHUNKS");
        if (e) {
            int want_c = atoi(e);
            if (want_c < 1) want_c = 1;
            if (want_c > 8) want_c = 8;
            while (want_c > 1 && !feasible(want_c)) --want_c;
            pl.nchunk = want_c;
        } else if (world > 1) {
            double best = 0.0;
            int best_c = 1;
            for (int c2 = 1; c2 <= 4; ++c2) {
                if (!feasible(c2)) continue;
                pl.nchunk = c2;
                plan_cost(pl, true);
                const double t = pl.model_comp + pl.model_wire;
                if (c2 == 1 || t < best) {
                    best = t;
                    best_c = c2;
                }
            }
            pl.nchunk = best_c;
        }
    }
    // ---- the cost model's choice (unless the caller forced one) ----
    plan_cost(pl, pl.cols);
    if (want == 0 && world > 1 && pl.model_comp + pl.model_wire >= 0.97 * pl.model_single) {
        const double ms1 = 1e3 * pl.model_single, msc = 1e3 * pl.model_comp, msw = 1e3 * pl.model_wire;
        const unsigned long long mb = pl.model_bytes;
        char why[160];
        snprintf(why, sizeof why, "cost model: %d ranks %.2f ms of work + %.2f ms on the wire (%.0f GB/s links) >= one GPU's %.2f ms", world, msc, msw,
                 link_gbs(), ms1);
        single(why);
        pl.model_single = 1e-3 * ms1;
        pl.model_comp = 1e-3 * msc;
        pl.model_wire = 1e-3 * msw;
        pl.model_bytes = mb;
        pl.cols = false;
    }
    // ---- plan 3 against whatever was chosen above (round 6): captures that have a multipole form (of the Hilbert transform and, where they are
    // resampled, of the resampler) take it where the model puts it 3 % or more ahead -- its exchanges are kilobytes, so it wins wherever the capture is long enough to be worth cutting at all ----
    if (want == 0 && world > 1 && !want_rows) {
        wfx_decode_params q = *p;
        q.shard_plan = 3;
        shard_plan f;
        wfx_fmm_shard_geo fgeo, rgeo;
        if (wfx_fmm_shard_geometry(p->n, &fgeo) == 0 && (!p->resample || wfx_rs_shard_geometry(p->n0, p->n, &rgeo) == 0) && make_plan(nullptr, &q, world, rank, f) == 0) {
            const double now = pl.single ? pl.model_single : pl.model_comp + pl.model_wire;
            if (f.model_comp + f.model_wire < 0.97 * now) {
                pl = f;
                pl.forced = 0;
            }
        }
    }
    return 0;
}

struct wfx_shard {
    wfx_ctx *ctx = nullptr;
    wfx_comm *comm = nullptr;
    wfx_decode_params dp{};
    shard_plan pl;
    wfx_dist dF, dI, dH;                  // resampler forward / inverse, Hilbert
    wfx_dist dHk;                         // padded form over several ranks: the kernel's transform (full rows), first decode only
    wfx_devbuf b_in, b_merged, b_res, b_audio, b_v, b_env, b_dig, b_blk, b_blks, b_nan, b_flags;
    wfx_devbuf b_gath, b_pieces;          // columns layout, rank 0: the ranks' segments as gathered, and the 2-D copies that put them in order
    int n_pieces = 0;
    long long piece_max = 0;
    wfx_devbuf b_grow, b_ghat;            // padded form: this rank's rows of the kernel g_ext / Kp, and its slab of the kernel's transform
    bool ghat_ready = false;              // computed by three extra phases in front of the first decode
    // those three phases have run but the decode they belonged to has not finished (ghat_ready selects the phase numbering, so it
    // cannot flip in the middle of one): promoted to ghat_ready when the next decode starts -- a decode that aborts in a later
    // phase and is retried starts from the audio phases, with the forward half already re-bound to the audio rows
    bool ghat_pending = false;
    bool in_decode = false;
    const void *ext_in = nullptr;
    bool have_input = false, ran = false, bound = false;
    uint64_t cap = SH_CAND_CAP;
    unsigned *ws = nullptr;
};

static void settle_kernel(wfx_shard *sh)
{
    if (sh->ghat_pending && !sh->in_decode) {
        sh->ghat_ready = true;
        sh->ghat_pending = false;
    }
}

static size_t frame_bytes(int in_kind) { return in_kind == WFX_IN_I16_MONO ? 2 : (in_kind == WFX_IN_I16_STEREO ? 4 : 8); }

static void free_buf(wfx_devbuf &b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

// ---- columns layout: buffers and bindings --------------------------------------------------------------------------------
// Audio (and, with a resampler, its input b_res) is R1 rows of xrs = 2 wH + 2 hs samples laid end to end, SH_HALO samples of
// slack in front and behind (the envelope kernel's clamped loads reach two samples beyond a row); the own samples of row q
// start at pad + q xrs + hs.  V: R1 rows of 2 + wH + 2 points.
#define SH_PAD SH_HALO
static uint64_t cols_own(const shard_plan &pl) { return (uint64_t)pl.g.R1 * 2ull * (uint64_t)pl.wH; }
static uint64_t cols_in_frames(const shard_plan &pl)
{
    return pl.resample ? (uint64_t)pl.g.R1 * 2ull * (uint64_t)pl.wF : (uint64_t)pl.g.R1 * (uint64_t)pl.xrs;
}

// Padded forms: a rank's segments reach past the capture's end.  Its stage buffers hold the slots of all R1 segments in order, so
// the slots that hold samples are a PREFIX: the segments in front of the row the capture ends in, and the part of that row's
// segment below n.  (Unpadded: every slot.)
static uint64_t cols_valid_of(const shard_plan &pl, int r)
{
    const long long c0 = (long long)r * pl.Ms / pl.world / 4 * 4, c1 = r + 1 == pl.world ? pl.Ms : (long long)(r + 1) * pl.Ms / pl.world / 4 * 4;
    const long long w = c1 - c0;
    if (!pl.padded) return (uint64_t)pl.g.R1 * 2ull * (uint64_t)w;
    const long long qe = (pl.K - 1) / pl.Ms;                       // the row that holds the capture's last point
    long long part = (long long)pl.n - 2 * (qe * pl.Ms + c0);
    part = part < 0 ? 0 : (part > 2 * w ? 2 * w : part);
    return (uint64_t)qe * 2ull * (uint64_t)w + (uint64_t)part;
}
static uint64_t cols_valid(const shard_plan &pl) { return cols_valid_of(pl, pl.rank); }

static int shard_bind_cols(wfx_shard *sh)
{
    wfx_ctx *ctx = sh->ctx;
    shard_plan &pl = sh->pl;
    const int me = pl.rank, W = pl.world, R1 = pl.g.R1;
    const uint64_t n_own = cols_own(pl), flat = (uint64_t)R1 * (uint64_t)pl.xrs;
    const void *in = sh->ext_in ? sh->ext_in : sh->b_in.p;
    WFX_TRY(wfx_reserve(ctx, sh->b_audio, (flat + 2 * SH_PAD) * 8 + 64));
    WFX_TRY(wfx_reserve(ctx, sh->b_v, (size_t)R1 * (size_t)(pl.wH + 2 * SH_VHALO) * sizeof(cplx) + 64));
    WFX_TRY(wfx_reserve(ctx, sh->b_env, n_own * 8 + 64));
    WFX_TRY(wfx_reserve(ctx, sh->b_dig, (me == 0 ? pl.n : n_own) + 64));
    if (me == 0 && W > 1) WFX_TRY(wfx_reserve(ctx, sh->b_gath, std::max((size_t)pl.n, (size_t)n_own) + 64 + 2 * (size_t)W));      // (n_own: wfx_shard_fetch reads all slots)
    WFX_TRY(wfx_reserve(ctx, sh->b_blk, wfx_select_block_bytes(sh->cap)));
    WFX_TRY(wfx_reserve(ctx, sh->b_blks, wfx_select_block_bytes(sh->cap) * W));
    WFX_TRY(wfx_reserve(ctx, sh->b_nan, 8 * (size_t)W + 64));
    WFX_TRY(wfx_reserve(ctx, sh->b_flags, 64));
    WFX_HIP(ctx, hipMemsetAsync(sh->b_flags.p, 0, 64, ctx->stream));
    WFX_HIP(ctx, hipMemsetAsync(sh->b_nan.p, 0, 8 * (size_t)W + 64, ctx->stream));
    if (!sh->bound) WFX_HIP(ctx, hipMemsetAsync(sh->b_audio.p, 0, (flat + 2 * SH_PAD) * 8 + 64, ctx->stream));
    double *audio_row0 = (double *)sh->b_audio.p + SH_PAD + pl.hs;          // first own sample of row 0
    if (pl.resample) {
        const void *cols_in = in;
        if (pl.in_kind == WFX_IN_I16_STEREO) {
            WFX_TRY(wfx_reserve(ctx, sh->b_merged, cols_in_frames(pl) * 8 + 64));
            cols_in = sh->b_merged.p;
        }
        WFX_TRY(wfx_reserve(ctx, sh->b_res, (flat + 2 * SH_PAD) * 8 + 64));
        if (!sh->bound) WFX_HIP(ctx, hipMemsetAsync(sh->b_res.p, 0, (flat + 2 * SH_PAD) * 8 + 64, ctx->stream));
        WFX_TRY(sh->dF.bind_cols(cols_in, pl.wF, nullptr, 0, 0));
        WFX_TRY(sh->dI.bind_cols(nullptr, 0, (cplx *)((double *)sh->b_res.p + SH_PAD), pl.xrs / 2, 0));
    } else if (pl.in_kind == WFX_IN_I16_STEREO) {
        WFX_TRY(wfx_reserve(ctx, sh->b_merged, flat * 8 + 64));
    }
    if (pl.padded) WFX_TRY(wfx_reserve(ctx, sh->b_ghat, (size_t)sh->dH.slab_points() * sizeof(cplx) + 64));
    // (until the kernel's transform exists the Hilbert transform's forward half reads the kernel's columns, which are written into
    // the V rows -- free until the first inverse pass; run_phase_cols re-binds it to the audio afterwards)
    if (pl.padded && !sh->ghat_ready)
        WFX_TRY(sh->dH.bind_cols((cplx *)sh->b_v.p + SH_VHALO, pl.wH + 2 * SH_VHALO, (cplx *)sh->b_v.p, pl.wH + 2 * SH_VHALO, sh->dH.fwd_result_index()));
    else
        WFX_TRY(sh->dH.bind_cols(audio_row0, pl.xrs / 2, (cplx *)sh->b_v.p, pl.wH + 2 * SH_VHALO, sh->dH.fwd_result_index()));
    // rank 0: the gathered segments of every rank -> the stream in order (2-byte elements: every offset is even)
    sh->n_pieces = 0;
    if (me == 0 && W > 1) {
        std::vector<wfx_dist_piece> ps;
        unsigned long long off = 0;
        sh->piece_max = 0;
        const long long qe = pl.padded ? (pl.K - 1) / pl.Ms : (long long)R1;      // rows in front of it are whole
        for (int r = 0; r < W; ++r) {
            const long long c0 = (long long)r * pl.Ms / W / 4 * 4, c1 = r + 1 == W ? pl.Ms : (long long)(r + 1) * pl.Ms / W / 4 * 4;
            const uint64_t nv = cols_valid_of(pl, r);
            wfx_dist_piece q{};
            q.src = (unsigned long long)((uint8_t *)sh->b_gath.p + off);
            q.dst = (unsigned long long)((uint8_t *)sh->b_dig.p + 2 * c0);
            q.rows = (int)(pl.padded ? qe : R1);
            q.cols = (int)(c1 - c0);                // 2-byte elements: one per packed point
            q.src_rs = c1 - c0;
            q.dst_rs = pl.Ms;
            if (q.rows > 0) {
                ps.push_back(q);
                sh->piece_max = std::max(sh->piece_max, (long long)q.rows * q.cols);
            }
            if (pl.padded) {      // the row the capture ends in: the part of the rank's segment below n (an odd count: one byte more, into the slack)
                const uint64_t part = nv - (uint64_t)qe * 2ull * (uint64_t)(c1 - c0);
                if (part) {
                    wfx_dist_piece e{};
                    e.src = q.src + (unsigned long long)qe * 2ull * (unsigned long long)(c1 - c0);
                    e.dst = (unsigned long long)((uint8_t *)sh->b_dig.p + 2 * (qe * pl.Ms + c0));
                    e.rows = 1;
                    e.cols = (int)((part + 1) / 2);
                    e.src_rs = e.cols;
                    e.dst_rs = e.cols;
                    ps.push_back(e);
                    sh->piece_max = std::max(sh->piece_max, (long long)e.cols);
                }
            }
            off += (nv + 1) & ~(uint64_t)1;          // (2-byte copies: every rank's bytes start on an even offset)
        }
        WFX_TRY(wfx_reserve(ctx, sh->b_pieces, ps.size() * sizeof(wfx_dist_piece) + 64));
        WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        WFX_HIP(ctx, hipMemcpy(sh->b_pieces.p, ps.data(), ps.size() * sizeof(wfx_dist_piece), hipMemcpyHostToDevice));
        sh->n_pieces = (int)ps.size();
    }
    sh->bound = true;
    return 0;
}

static int shard_bind(wfx_shard *sh)
{
    wfx_ctx *ctx = sh->ctx;
    shard_plan &pl = sh->pl;
    settle_kernel(sh);
    if (pl.single) {                      // rank 0 decodes alone (the fused path owns its buffers)
        sh->bound = true;
        return 0;
    }
    if (pl.cols) return shard_bind_cols(sh);
    if (pl.fmm) {
        const int me = pl.rank;
        const uint64_t n_own = pl.own_hi - pl.own_lo;
        WFX_TRY(wfx_reserve(ctx, sh->b_audio, (n_own + 2 * SH_FMM_HF) * 8 + 64));
        WFX_TRY(wfx_reserve(ctx, sh->b_res, 2 * (size_t)(SH_FMM_HR + 24) * 8 + 64));          // the two halo segments as the notch leaves them
        if (pl.rs) {
            // the rank's input samples as float64 (unless they arrive so), and its resampled samples with 320 beyond either end
            if (pl.in_kind != WFX_IN_F64_MONO) WFX_TRY(wfx_reserve(ctx, sh->b_merged, (pl.in_hi - pl.in_lo + 2 * SH_RS_HS) * 8 + 64));
            WFX_TRY(wfx_reserve(ctx, sh->b_v, (n_own + 2 * SH_FMM_HR) * 8 + 64));
        } else if (pl.in_kind == WFX_IN_I16_STEREO)
            WFX_TRY(wfx_reserve(ctx, sh->b_merged, (n_own + 2 * SH_FMM_HR) * 8 + 64));
        WFX_TRY(wfx_reserve(ctx, sh->b_env, n_own * 8 + 64));
        WFX_TRY(wfx_reserve(ctx, sh->b_dig, (me == 0 ? pl.n : n_own) + 64));
        WFX_TRY(wfx_reserve(ctx, sh->b_blk, wfx_select_block_bytes(sh->cap)));
        WFX_TRY(wfx_reserve(ctx, sh->b_blks, wfx_select_block_bytes(sh->cap) * pl.world));
        WFX_TRY(wfx_reserve(ctx, sh->b_nan, 8 * (size_t)pl.world + 64));
        WFX_TRY(wfx_reserve(ctx, sh->b_flags, 64));
        WFX_HIP(ctx, hipMemsetAsync(sh->b_flags.p, 0, 64, ctx->stream));
        WFX_HIP(ctx, hipMemsetAsync(sh->b_nan.p, 0, 8 * (size_t)pl.world + 64, ctx->stream));
        sh->bound = true;
        return 0;
    }
    const int me = pl.rank;
    const long long nr = pl.g.nrows(me);
    const uint64_t n_own = pl.own_hi - pl.own_lo, n_seg = pl.seg_hi - pl.seg_lo;
    const void *in = sh->ext_in ? sh->ext_in : sh->b_in.p;
    // audio segment [seg_lo, seg_hi): the notch output; the Hilbert transform's input rows start at own_lo
    // (padded form: the Hilbert transform reads all nr Ms points of the rank's rows starting at its first own sample; what lies
    // beyond the capture is the zero padding -- the notch never writes there, one memset at binding time is enough)
    const size_t audio_bytes = pl.padded ? std::max((size_t)n_seg, (size_t)(pl.own_lo - pl.seg_lo) + (size_t)pl.spp * (size_t)nr * (size_t)pl.Ms) * 8 + 64 : n_seg * 8 + 64;
    const bool audio_moved = !sh->b_audio.p || sh->b_audio.cap < audio_bytes;
    WFX_TRY(wfx_reserve(ctx, sh->b_audio, audio_bytes));
    if (pl.padded && (audio_moved || !sh->bound)) WFX_HIP(ctx, hipMemsetAsync(sh->b_audio.p, 0, audio_bytes, ctx->stream));
    WFX_TRY(wfx_reserve(ctx, sh->b_v, (size_t)(2 * SH_VHALO + nr * pl.Ms) * sizeof(cplx) + 64));
    if (pl.padded) {
        if (!sh->ghat_ready) WFX_TRY(wfx_reserve(ctx, sh->b_grow, (size_t)(pl.split_kernel ? pl.gk.nrows(me) : nr) * pl.Ms * sizeof(cplx) + 64));
        WFX_TRY(wfx_reserve(ctx, sh->b_ghat, (size_t)sh->dH.slab_points() * sizeof(cplx) + 64));
    }
    WFX_TRY(wfx_reserve(ctx, sh->b_env, n_own * 8 + 64));
    WFX_TRY(wfx_reserve(ctx, sh->b_dig, (me == 0 ? pl.n : n_own) + 64));
    WFX_TRY(wfx_reserve(ctx, sh->b_blk, wfx_select_block_bytes(sh->cap)));
    WFX_TRY(wfx_reserve(ctx, sh->b_blks, wfx_select_block_bytes(sh->cap) * pl.world));
    WFX_TRY(wfx_reserve(ctx, sh->b_nan, 8 * (size_t)pl.world + 64));
    WFX_TRY(wfx_reserve(ctx, sh->b_flags, 64));
    WFX_HIP(ctx, hipMemsetAsync(sh->b_flags.p, 0, 64, ctx->stream));
    WFX_HIP(ctx, hipMemsetAsync(sh->b_nan.p, 0, 8 * (size_t)pl.world + 64, ctx->stream));
    const double *audio_own = (const double *)sh->b_audio.p + (pl.own_lo - pl.seg_lo);
    if (pl.resample) {
        const void *rows_in = in;
        if (pl.in_kind == WFX_IN_I16_STEREO) {
            WFX_TRY(wfx_reserve(ctx, sh->b_merged, (pl.in_hi - pl.in_lo) * 8 + 64));
            rows_in = sh->b_merged.p;
        }
        // resampled audio: [own_lo - 32, own_hi + 32) circularly (16 points of halo on either side)
        WFX_TRY(wfx_reserve(ctx, sh->b_res, (size_t)(SH_HALO + nr * pl.Ms) * sizeof(cplx) + 64));
        WFX_TRY(sh->dF.bind(rows_in, nullptr, 0));
        WFX_TRY(sh->dI.bind(nullptr, (cplx *)sh->b_res.p, 0));
    } else if (pl.in_kind == WFX_IN_I16_STEREO) {
        WFX_TRY(wfx_reserve(ctx, sh->b_merged, (pl.in_hi - pl.in_lo) * 8 + 64));
    }
    // (until the kernel's transform exists the Hilbert transform's forward half is bound to the kernel's rows: see run_phase;
    // with the rows split, the kernel has a transform object of its own)
    if (pl.split_kernel && !sh->ghat_ready) WFX_TRY(sh->dHk.bind(sh->b_grow.p, nullptr, 0));
    WFX_TRY(sh->dH.bind((pl.padded && !pl.split_kernel && !sh->ghat_ready) ? (const void *)sh->b_grow.p : (const void *)audio_own, (cplx *)sh->b_v.p,
                        sh->dH.fwd_result_index()));
    sh->bound = true;
    return 0;
}

// ---- the phases ------------------------------------------------------------------------------------------------
// padded form: + 1 phase (the wrap of V, see phase 8) and, in front of the first decode, + 3 (the kernel's transform)
static int phase_count(const wfx_shard *sh)
{
    if (sh->pl.single) return 1;
    if (sh->pl.fmm) return sh->pl.rs ? 9 : 7;
    if (sh->pl.cols)      // padded: + 1 (V[K] := V[0], even lengths) and, in front of the first decode, + 2 C (the kernel's transform)
        return (sh->pl.resample ? 4 * sh->pl.nchunk + 7 : 2 * sh->pl.nchunk + 6) + (sh->pl.padded && !sh->pl.plain ? 1 : 0) +
               (sh->pl.padded && !sh->ghat_ready ? 2 * sh->pl.nchunk : 0);
    return sh->pl.resample ? 13 : (sh->pl.padded ? (sh->ghat_ready ? 10 : 13) : 9);
}

// ---- the phases of the columns layout -------------------------------------------------------------------------------
static int run_phase_cols(wfx_shard *sh, int ph)
{
    wfx_ctx *ctx = sh->ctx;
    wfx_comm *c = sh->comm;
    shard_plan &pl = sh->pl;
    const wfx_decode_params &p = sh->dp;
    const int me = pl.rank, W = pl.world, R1 = pl.g.R1;
    const uint64_t n_own = cols_own(pl), flat = (uint64_t)R1 * (uint64_t)pl.xrs;
    wfx_dev_scalars *ds = (wfx_dev_scalars *)ctx->b_scal.p;
    const void *in = sh->ext_in ? sh->ext_in : sh->b_in.p;
    double *audio = (double *)sh->b_audio.p + SH_PAD;                       // row 0, halo included
    double *env = (double *)sh->b_env.p;
    uint8_t *dig_own = (me == 0 && W > 1) ? (uint8_t *)sh->b_gath.p : (uint8_t *)sh->b_dig.p;
    // Phases.  C = k1 subsets per rank.  A distributed transform pair takes 2 C + 1 phases:
    //   [0]            first pass on the columns, E2 of subset 0          (exchanges are asynchronous on RCCL: slot s)
    //   [1 .. C-1]     E2 of subset c
    //   [C + c]        wait E2(c); slab passes of subset c (forward, spectral step, inverse); E3 of subset c
    //   [2 C]          wait every E3; last pass; halo exchange
    // the resampler's pair first (when there is one), then the Hilbert transform's, then the five phases of the tail.
    const int C = pl.nchunk, TP = 2 * C + 1;
    const uint64_t n_valid = cols_valid(pl);                               // slots of the stage buffers that hold samples (a prefix)
    const long long vrs = pl.wH + 2 * SH_VHALO;
    if (pl.padded && !sh->ghat_ready) {
        // ---- the padded convolution's kernel, transformed once per shard, in columns: this rank's columns of the kernel's rows
        // (written into the V rows) -> first pass -> E2 subset by subset -> slab passes -> the table, subset by subset ----
        if (ph < 2 * C) {
            if (ph == 0) {
                cplx *vrow0 = (cplx *)sh->b_v.p + SH_VHALO;
                for (int q = 0; q < R1; ++q) {
                    const long long p0 = (long long)q * pl.Ms + pl.cH0;
                    if (pl.plain)
                        WFX_TRY(wfx_dev_hilbert_kernel_rows_real(ctx, vrow0 + (long long)q * vrs, p0, pl.wH, (long long)pl.n, pl.Kp));
                    else
                        WFX_TRY(wfx_dev_hilbert_kernel_rows(ctx, vrow0 + (long long)q * vrs, p0, pl.wH, (long long)pl.n, pl.Kp));
                }
                WFX_TRY(sh->dH.fwd_pass1(0));
                return sh->dH.e2_exchange(c, 0, 0);
            }
            if (ph < C) return sh->dH.e2_exchange(c, ph, ph);
            const int k = ph - C;
            WFX_TRY(wfx_comm_wait(c, ctx, k));
            cplx *G = nullptr;
            WFX_TRY(sh->dH.fwd_slab_chunk(k, 0, &G));
            const long long off = sh->dH.chunk_offset(k), cnt = sh->dH.chunk_offset(k + 1) - off;
            if (pl.plain)      // the real kernel's table: one double per slab entry, untangled from its packed transform
                WFX_TRY(wfx_dist_real_untangle_km(ctx, R1, sh->dH.chunk_kmap(k), G, pl.Kp, (double *)sh->b_ghat.p + off));
            else
                WFX_HIP(ctx, hipMemcpyAsync((cplx *)sh->b_ghat.p + off, G, (size_t)cnt * sizeof(cplx), hipMemcpyDeviceToDevice, ctx->stream));
            if (k == C - 1) {
                // from here on the forward half reads the audio rows (re-binding rebuilds the lists: a host synchronisation, once)
                WFX_TRY(sh->dH.bind_cols(audio + pl.hs, pl.xrs / 2, (cplx *)sh->b_v.p, vrs, sh->dH.fwd_result_index()));
                sh->ghat_pending = true;
            }
            return 0;
        }
        ph -= 2 * C;
    }
    if (!pl.resample) ph += TP;
    if (ph < TP) {
        // ---- a4 + a5: distributed rfft -> scipy.signal.resample's bin copy -> distributed irfft, in columns ----
        if (ph == 0) {
            WFX_HIP(ctx, hipMemsetAsync(ds, 0, sizeof(wfx_dev_scalars), ctx->stream));
            if (pl.in_kind == WFX_IN_I16_STEREO) WFX_TRY(wfx_dev_merge(ctx, (const int16_t *)in, cols_in_frames(pl), (double *)sh->b_merged.p));
            WFX_TRY(sh->dF.fwd_pass1(pl.in_kind == WFX_IN_I16_MONO ? 2 : 0));
            return sh->dF.e2_exchange(c, 0, 0);
        }
        if (ph < C) return sh->dF.e2_exchange(c, ph, ph);
        if (ph < 2 * C) {
            const int k = ph - C;
            WFX_TRY(wfx_comm_wait(c, ctx, k));
            cplx *Z = nullptr;
            // down-sampling reads the bins [0, n/2] and their mirrors only: the last forward pass does not store the rest
            const long long nmin = (long long)(pl.n0 < pl.n ? pl.n0 : pl.n), half = nmin / 2;
            WFX_TRY(sh->dF.fwd_slab_chunk(k, 0, &Z, half, pl.M1 - half));
            WFX_TRY(wfx_dist_resample_glue_km(ctx, pl.g.R1, sh->dF.chunk_kmap(k), Z, (long long)pl.n0, (long long)pl.n, sh->dI.slab_chunk(0, k)));
            WFX_TRY(sh->dI.inv_slab_chunk(k, sh->dI.slab_chunk(0, k)));
            return sh->dI.e3_exchange(c, k, C + k);
        }
        for (int k = 0; k < C; ++k) WFX_TRY(wfx_comm_wait(c, ctx, C + k));
        return sh->dI.inv_pass1_halo_exchange(c);                       // the resampled audio, in columns; its halo columns
    }
    ph -= TP;
    if (ph < TP) {
        if (ph == 0) {   // a6 notch over the rows laid end to end, then the Hilbert transform's first pass (in place) and E2
            const void *nin = in;
            int nkind = pl.in_kind;
            if (pl.resample) {
                WFX_TRY(sh->dI.inv_halo_unpack());
                nin = (const double *)sh->b_res.p + SH_PAD;
                nkind = WFX_IN_F64_MONO;
            } else {
                WFX_HIP(ctx, hipMemsetAsync(ds, 0, sizeof(wfx_dev_scalars), ctx->stream));
                if (pl.in_kind == WFX_IN_I16_STEREO) {
                    WFX_TRY(wfx_dev_merge(ctx, (const int16_t *)in, flat, (double *)sh->b_merged.p));
                    nin = sh->b_merged.p;
                    nkind = WFX_IN_F64_MONO;
                }
            }
            // rank 0's first row starts at the capture's true start, the last rank's last row ends at its true end: filtfilt's exact
            // edges there; everywhere else the 49-tap form, whose outputs within 24 samples of a row's end are never read
            uint64_t lo = me == 0 ? (uint64_t)pl.hs : 0, hi = flat - (me == W - 1 ? (uint64_t)pl.hs : 0);
            int flags = (me == 0 ? 1 : 0) | (me == W - 1 ? 2 : 0);
            if (pl.padded) {
                // The capture ends inside row qe of the arrangement: at position e of this rank's segment of that row (left halo
                // included), if it sees it.  The filter runs up to there -- with filtfilt's exact edge when the end lies at or
                // behind the rank's own columns (its 127-sample history is then inside the segment: hs = 192) -- and not at all
                // behind it: what follows in the audio rows is the convolution's zero padding, cleared when they were bound.
                const long long qe = (pl.K - 1) / pl.Ms;
                const long long e = (long long)pl.n - (2 * (qe * pl.Ms + pl.cH0) - pl.hs);
                flags &= 1;
                // (the LAST rank's segment of row qe - 1 ends with row qe's first columns as its right halo: a capture that ends up to
                // 63 samples behind a row boundary has part of filtfilt's edge -- the last 64 outputs -- among this rank's own samples
                // of row qe - 1, and the edge is then applied where the capture ends inside that halo)
                const long long e1 = qe > 0 ? e + 2 * pl.Ms : -1;        // the end relative to this rank's segment of row qe - 1
                if (e < pl.hs && me == W - 1 && e1 >= pl.hs && e1 <= pl.xrs) {
                    hi = (uint64_t)(qe - 1) * (uint64_t)pl.xrs + (uint64_t)e1;
                    flags |= 2;
                } else if (e < pl.hs)
                    hi = (uint64_t)qe * (uint64_t)pl.xrs;                 // nothing of row qe is this rank's
                else if (e <= pl.xrs) {
                    hi = (uint64_t)qe * (uint64_t)pl.xrs + (uint64_t)e;
                    flags |= 2;
                } else
                    hi = (uint64_t)(qe + 1) * (uint64_t)pl.xrs;
            }
            double ext18[18];
            const bool use_ext = p.has_ext && !pl.resample && pl.in_kind != WFX_IN_I16_STEREO;
            for (int i = 0; i < 9; ++i) {
                ext18[i] = p.ext_left[i];
                ext18[9 + i] = p.ext_right[i];
            }
            const void *nin_lo = nkind == WFX_IN_I16_MONO ? (const void *)((const int16_t *)nin + lo) : (const void *)((const double *)nin + lo);
            if (hi > lo) WFX_TRY(wfx_dev_notch_fir_only(ctx, nin_lo, nkind, hi - lo, p.notch_b, p.notch_a, audio + lo, flags, use_ext ? ext18 : nullptr));
            WFX_TRY(sh->dH.fwd_pass1(1));
            return sh->dH.e2_exchange(c, 0, 2 * C);
        }
        if (ph < C) return sh->dH.e2_exchange(c, ph, 2 * C + ph);
        if (ph < 2 * C) {
            const int k = ph - C;
            WFX_TRY(wfx_comm_wait(c, ctx, 2 * C + k));
            cplx *G = nullptr;
            const long long goff = sh->dH.chunk_offset(k);
            if (pl.plain) {       // odd length: real samples, real kernel -- the glue between the packed forward and inverse transforms
                WFX_TRY(sh->dH.fwd_slab_chunk(k, 0, &G));
                WFX_TRY(wfx_dist_real_conv_glue_km(ctx, R1, sh->dH.chunk_kmap(k), G, pl.Kp, (const double *)sh->b_ghat.p + goff));
            } else if (pl.padded)
                WFX_TRY(sh->dH.fwd_slab_chunk(k, 0, &G, 0, 0, (const cplx *)sh->b_ghat.p + goff));
            else
                WFX_TRY(sh->dH.fwd_slab_chunk(k, 1, &G));
            WFX_TRY(sh->dH.inv_slab_chunk(k, G));
            return sh->dH.e3_exchange(c, k, 3 * C + k);
        }
        for (int k = 0; k < C; ++k) WFX_TRY(wfx_comm_wait(c, ctx, 3 * C + k));
        return sh->dH.inv_pass1_halo_exchange(c);
    }
    ph -= TP;
    const bool wrap = pl.padded && !pl.plain;
    if (wrap) {
        if (ph == 0) {
            // even padded form: the cyclic result's V[K] is V[0] (H[n - 1] sits in its .y) -- it lives in rank 0's first column, the
            // rank that holds the capture's last point needs it in the slot behind that point (its next column, or its right halo)
            WFX_TRY(sh->dH.inv_halo_unpack());
            const long long q1 = (pl.K - 1) / pl.Ms, c1 = (pl.K - 1) - q1 * pl.Ms;
            int rk = 0;
            long long ck0 = 0;
            for (int r = 0; r < W; ++r) {
                const long long a0 = (long long)r * pl.Ms / W / 4 * 4, a1 = r + 1 == W ? pl.Ms : (long long)(r + 1) * pl.Ms / W / 4 * 4;
                if (c1 >= a0 && c1 < a1) {
                    rk = r;
                    ck0 = a0;
                }
            }
            wfx_xfer x{};
            x.peer = me == 0 ? rk : 0;
            if (me == 0) {
                x.send = (cplx *)sh->b_v.p + SH_VHALO;
                x.send_bytes = sizeof(cplx);
            }
            if (me == rk) {
                x.recv = (cplx *)sh->b_v.p + q1 * vrs + SH_VHALO + (c1 - ck0) + 1;
                x.recv_bytes = sizeof(cplx);
            }
            wfx_comm_label(c, "hilbert wrap");
            return (x.send_bytes || x.recv_bytes) ? wfx_comm_exchange(c, ctx, &x, 1) : wfx_comm_exchange(c, ctx, &x, 0);
        }
        ph -= 1;
    }
    ph += 6;
    switch (ph) {
    case 6: {   // a7 envelope + median per segment, level-0 histogram; first all-reduce
        if (!wrap) WFX_TRY(sh->dH.inv_halo_unpack());                          // (even padded form: done with the wrap, one phase earlier)
        WFX_TRY(wfx_dev_select_sharded_ws(ctx, &sh->ws));
        WFX_TRY(wfx_dev_env_median_segs(ctx, (const cplx *)sh->b_v.p + SH_VHALO, pl.wH + 2 * SH_VHALO, audio + pl.hs, pl.xrs, R1, (int)(2 * pl.wH),
                                        2 * pl.cH0, 2 * pl.Ms, pl.n, env, sh->ws, pl.plain ? 1 : 0));
        wfx_comm_label(c, "select level 0");
        return wfx_comm_allreduce_u32(c, ctx, sh->ws, WFX_SEL_BINS);
    }
    case 7: {
        const uint64_t ranks[4] = {p.rank_lo[0], p.rank_lo[1], p.rank_hi[0], p.rank_hi[1]};
        WFX_TRY(wfx_dev_select_l1(ctx, env, n_valid, ranks, sh->ws, ds));
        wfx_comm_label(c, "select level 1");
        return wfx_comm_allreduce_u32(c, ctx, sh->ws + WFX_SEL_H1_OFFSET, WFX_SEL_H1_WORDS);
    }
    case 8: {
        WFX_TRY(wfx_dev_select_compact_block(ctx, env, n_valid, sh->ws, ds, sh->b_blk.p, sh->cap));
        wfx_comm_label(c, "select candidates");
        return wfx_comm_allgather(c, ctx, sh->b_blk.p, sh->b_blks.p, wfx_select_block_bytes(sh->cap));
    }
    case 9: {   // a8 finish + quantise; the one gather of the stream (every rank's segments, back to back)
        WFX_TRY(wfx_dev_select_finish_blocks(ctx, sh->ws, ds, sh->b_blks.p, W, sh->cap, p.gamma_lo, p.gamma_hi, (unsigned *)sh->b_flags.p));
        if (n_valid) WFX_TRY(wfx_dev_quantise(ctx, env, n_valid, ds, dig_own, ds));
        std::vector<wfx_xfer> xs;
        if (me == 0) {
            uint64_t off = (n_valid + 1) & ~(uint64_t)1;                      // (as shard_bind_cols lays the pieces out)
            for (int s = 1; s < W; ++s) {
                const uint64_t nb = cols_valid_of(pl, s);
                xs.push_back(wfx_xfer{s, nullptr, 0, (uint8_t *)sh->b_gath.p + off, (size_t)nb});
                xs.push_back(wfx_xfer{s, nullptr, 0, (unsigned long long *)sh->b_nan.p + s, 8});
                off += (nb + 1) & ~(uint64_t)1;
            }
        } else {
            xs.push_back(wfx_xfer{0, dig_own, (size_t)n_valid, nullptr, 0});
            xs.push_back(wfx_xfer{0, &ds->nan_count, 8, nullptr, 0});
        }
        wfx_comm_label(c, "stream gather");
        return wfx_comm_exchange(c, ctx, xs.data(), (int)xs.size());
    }
    case 10: {  // rank 0: the segments in order, then a9 + a10
        if (me != 0) return 0;
        if (W > 1) {
            WFX_TRY(wfx_dist_copy2d(ctx, (const wfx_dist_piece *)sh->b_pieces.p, sh->n_pieces, sh->piece_max, 2));
            WFX_TRY(wfx_dev_add_u64(ctx, &ds->nan_count, (const unsigned long long *)sh->b_nan.p + 1, W - 1));
        }
        const int w = p.width;
        const int h_max = (int)(pl.n / (uint64_t)w);
        WFX_TRY(wfx_reserve(ctx, ctx->b_img, (size_t)w * 4 * (size_t)(h_max > 0 ? h_max : 1)));
        WFX_TRY(wfx_dev_sync_pick(ctx, (const uint8_t *)sh->b_dig.p, pl.n, p.n1, p.n0_gap, p.mindistance, p.frame_samples, w, ds));
        return wfx_dev_image(ctx, (const uint8_t *)sh->b_dig.p, pl.n, w, h_max, ds, (uint8_t *)ctx->b_img.p, ctx->h_scal);
    }
    default: return wfx_fail(ctx, WFX_ERR_BAD_ARG, "no such phase");
    }
}

// ---- plan 3: the phases in front of the select (wfx_fmm.hip has the kernels) ---------------------------------------
// the boxes of level lg (>= the level the ranks are dealt boxes of) that rank r owns
static void fmm_rank_boxes_lv(const shard_plan &pl, int lg, int r, long long *lo, long long *hi)
{
    const long long G = 1ll << pl.lgc;
    *lo = ((long long)r * G / pl.world) << (lg - pl.lgc);
    *hi = ((long long)(r + 1) * G / pl.world) << (lg - pl.lgc);
}

static void fmm_rank_boxes(const shard_plan &pl, int r, long long *lo, long long *hi) { fmm_rank_boxes_lv(pl, pl.fg.lg, r, lo, hi); }

static void fmm_rank_range(const shard_plan &pl, int r, uint64_t *lo, uint64_t *hi)
{
    long long a, b;
    fmm_rank_boxes(pl, r, &a, &b);
    const int sh = pl.fg.ltop - pl.fg.lg + 6;
    *lo = (uint64_t)wfx_fmm_leaf_first_host(pl.n, pl.fg.L, a << sh);
    *hi = (uint64_t)wfx_fmm_leaf_first_host(pl.n, pl.fg.L, b << sh);
}

static int fmm_owner_lv(const shard_plan &pl, int lg, long long gbox)          // the rank that owns box `gbox` of level lg
{
    const long long G = 1ll << pl.lgc, cb = gbox >> (lg - pl.lgc);
    int r = (int)((cb * pl.world + pl.world - 1) / G);               // a first guess, then settle
    if (r >= pl.world) r = pl.world - 1;
    for (;;) {
        const long long a = (long long)r * G / pl.world, b = (long long)(r + 1) * G / pl.world;
        if (cb < a)
            --r;
        else if (cb >= b)
            ++r;
        else
            return r;
    }
}

// what travels after the upward pass, as (sender, receiver, level, first box, boxes) in ONE order every rank enumerates alike: the gather level's
// boxes of every rank to every other rank, then per finer level and receiver the three boxes before its range and the three behind it
template <typename F>
static void fmm_weight_messages_lv(const shard_plan &pl, int lg, int L, F &&f)
{
    const int W = pl.world;
    for (int s = 0; s < W; ++s)
        for (int r = 0; r < W; ++r) {
            if (s == r) continue;
            long long a, b;
            fmm_rank_boxes_lv(pl, lg, s, &a, &b);
            f(s, r, lg, a, b - a);
        }
    for (int lev = lg + 1; lev <= L; ++lev) {
        const long long nbl = 1ll << lev;
        for (int r = 0; r < W; ++r) {
            long long a, b;
            fmm_rank_boxes_lv(pl, lg, r, &a, &b);
            const long long lo = a << (lev - lg), hi = b << (lev - lg);
            for (int side = 0; side < 2; ++side) {
                // the three boxes of a side travel as ONE message where they are neighbours in memory and have one owner (a rank owns at
                // least two boxes of every finer level, and the circle closes only between the last box and the first)
                long long run_box = -1, run_cnt = 0;
                int run_s = -1;
                for (int j = 0; j < 3; ++j) {
                    const long long raw = side == 0 ? lo - 3 + j : hi + j;
                    const long long box = (raw + nbl) & (nbl - 1);
                    const bool own = box >= lo && box < hi;                 // (one rank, or a range that is the whole level: its own)
                    const int s = own ? r : fmm_owner_lv(pl, lg, box >> (lev - lg));
                    if (run_cnt && (own || s != run_s || box != run_box + run_cnt)) {
                        f(run_s, r, lev, run_box, run_cnt);
                        run_cnt = 0;
                    }
                    if (own) continue;
                    if (!run_cnt) {
                        run_box = box;
                        run_s = s;
                    }
                    ++run_cnt;
                }
                if (run_cnt) f(run_s, r, lev, run_box, run_cnt);
            }
        }
    }
}

template <typename F>
static void fmm_weight_messages(const shard_plan &pl, F &&f)
{
    fmm_weight_messages_lv(pl, pl.fg.lg, pl.fg.L, f);
}

// plan 3 in front of a resampler: the two phases that leave the rank's resampled samples (and 320 of either neighbour's) where the phases of
// run_phase_fmm expect a float64 capture at 11 025 Hz
static int run_phase_rs(wfx_shard *sh, int ph)
{
    wfx_ctx *ctx = sh->ctx;
    wfx_comm *c = sh->comm;
    shard_plan &pl = sh->pl;
    const int me = pl.rank, W = pl.world;
    const uint64_t n_own = pl.own_hi - pl.own_lo, n_src = pl.in_hi - pl.in_lo;
    const void *in = sh->ext_in ? sh->ext_in : sh->b_in.p;
    const double *x = pl.in_kind == WFX_IN_F64_MONO ? (const double *)in : (const double *)sh->b_merged.p;
    const long long x0 = (long long)pl.in_lo - SH_RS_HS;
    double *y = (double *)sh->b_v.p;
    const uint64_t nkey = pl.n0 + (pl.n0 & 1);
    if (ph == 0) {
        if (pl.in_kind == WFX_IN_I16_STEREO)
            WFX_TRY(wfx_dev_merge(ctx, (const int16_t *)in, n_src + 2 * SH_RS_HS, (double *)sh->b_merged.p));
        else if (pl.in_kind == WFX_IN_I16_MONO)
            WFX_TRY(wfx_dev_i16_to_f64(ctx, (const int16_t *)in, n_src + 2 * SH_RS_HS, (double *)sh->b_merged.p));
        double *gsum = nullptr;
        WFX_TRY(wfx_rs_shard_up(ctx, x, x0, pl.n0, pl.n, pl.gbr_lo, pl.gbr_hi, &gsum));
        std::vector<wfx_xfer> xs;
        int rc = 0;
        fmm_weight_messages_lv(pl, pl.fgr.lg, pl.fgr.L, [&](int s, int r, int lev, long long box, long long cnt) {
            if (rc != 0 || (s != me && r != me)) return;
            double *ptr = nullptr;
            rc = wfx_fmm_shard_weights(ctx, nkey, lev, box, &ptr);
            wfx_xfer xf{};
            xf.peer = s == me ? r : s;
            if (s == me) {
                xf.send = ptr;
                xf.send_bytes = (size_t)cnt * 256;
            } else {
                xf.recv = ptr;
                xf.recv_bytes = (size_t)cnt * 256;
            }
            xs.push_back(xf);
        });
        WFX_TRY(rc);
        // the parts of C = sum x_n cos(..): one number per box of the resampler's gather level, every rank's to every other rank
        for (int s = 0; s < W; ++s)
            for (int r = 0; r < W; ++r) {
                if (s == r || (s != me && r != me)) continue;
                long long a, b;
                fmm_rank_boxes_lv(pl, pl.fgr.lg, s, &a, &b);
                wfx_xfer xf{};
                xf.peer = s == me ? r : s;
                if (s == me) {
                    xf.send = gsum + a;
                    xf.send_bytes = (size_t)(b - a) * 8;
                } else {
                    xf.recv = gsum + a;
                    xf.recv_bytes = (size_t)(b - a) * 8;
                }
                xs.push_back(xf);
            }
        wfx_comm_label(c, "resampler weights");
        return wfx_comm_exchange(c, ctx, xs.data(), (int)xs.size());
    }
    // ph == 1: the rest of the tree, the rank's targets; then its first 320 resampled samples to the rank before it, its last 320 to the one
    // behind it (round the circle: the capture's ends meet)
    WFX_TRY(wfx_rs_shard_down(ctx, x, x0, pl.n0, pl.n, pl.gbr_lo, pl.gbr_hi, y, (long long)pl.own_lo - SH_FMM_HR));
    std::vector<wfx_xfer> xs;
    double *own = y + SH_FMM_HR;
    for (int dir = 0; dir < 2; ++dir)
        for (int s = 0; s < W; ++s) {
            const int r = dir == 0 ? (s + 1) % W : (s + W - 1) % W;
            if (s != me && r != me) continue;
            // dir 0: s's last 320 -> the 320 in front of r's own; dir 1: s's first 320 -> the 320 behind r's own
            double *src = dir == 0 ? own + n_own - SH_FMM_HR : own, *dst = dir == 0 ? y : own + n_own;
            if (s == r) {
                WFX_HIP(ctx, hipMemcpyAsync(dst, src, SH_FMM_HR * 8, hipMemcpyDeviceToDevice, ctx->stream));
                continue;
            }
            wfx_xfer xf{};
            xf.peer = s == me ? r : s;
            if (s == me) {
                xf.send = src;
                xf.send_bytes = SH_FMM_HR * 8;
            } else {
                xf.recv = dst;
                xf.recv_bytes = SH_FMM_HR * 8;
            }
            xs.push_back(xf);
        }
    wfx_comm_label(c, "resampled halos");
    return wfx_comm_exchange(c, ctx, xs.data(), (int)xs.size());
}

static int run_phase_fmm(wfx_shard *sh, int ph)
{
    wfx_ctx *ctx = sh->ctx;
    wfx_comm *c = sh->comm;
    shard_plan &pl = sh->pl;
    const wfx_decode_params &p = sh->dp;
    const int me = pl.rank, W = pl.world;
    const uint64_t n_own = pl.own_hi - pl.own_lo;
    wfx_dev_scalars *ds = (wfx_dev_scalars *)ctx->b_scal.p;
    // (behind the resampler: its output, float64, [own_lo - 320, own_hi + 320) as run_phase_rs left it)
    const void *in = pl.rs ? sh->b_v.p : (sh->ext_in ? sh->ext_in : sh->b_in.p);
    const int in_kind = pl.rs ? WFX_IN_F64_MONO : pl.in_kind;
    double *audio = (double *)sh->b_audio.p;
    const long long audio0 = (long long)pl.own_lo - SH_FMM_HF, raw0 = (long long)pl.own_lo - SH_FMM_HR;
    double ext18[18];
    const bool use_ext = p.has_ext && in_kind != WFX_IN_I16_STEREO && !pl.rs;
    for (int i = 0; i < 9; ++i) {
        ext18[i] = p.ext_left[i];
        ext18[9 + i] = p.ext_right[i];
    }
    if (ph == 0) {
        // the rank's frames: [own_lo - 320, own_hi + 320) round the circle.  a4, then the leaf before the first own sample and the one behind the
        // last through the notch (segment form; at the capture's ends filtfilt's exact edges), then notch + P2M + M2M of the own workgroups
        int kind = in_kind;
        if (in_kind == WFX_IN_I16_STEREO) {
            WFX_TRY(wfx_dev_merge(ctx, (const int16_t *)in, n_own + 2 * SH_FMM_HR, (double *)sh->b_merged.p));
            in = sh->b_merged.p;
            kind = WFX_IN_F64_MONO;
        }
        const size_t fb = kind == WFX_IN_I16_MONO ? 2 : 8;
        double *tl = (double *)sh->b_res.p, *tr = tl + SH_FMM_HR + 24;
        const bool at_start = pl.own_lo == 0, at_end = pl.own_hi == pl.n;
        // left: the samples [own_lo - 320, own_lo) -- for the rank that holds the capture's start they are its END (exact right edge)
        WFX_TRY(wfx_dev_notch_fir_only(ctx, in, kind, at_start ? SH_FMM_HR : SH_FMM_HR + 24, p.notch_b, p.notch_a, tl, at_start ? 2 : 0, use_ext ? ext18 : nullptr));
        WFX_HIP(ctx, hipMemcpyAsync(audio, tl + 24, SH_FMM_HF * 8, hipMemcpyDeviceToDevice, ctx->stream));
        // right: [own_hi, own_hi + 320) -- for the rank that holds the capture's end they are its START (exact left edge)
        const char *rin = (const char *)in + (size_t)(SH_FMM_HR + n_own - (at_end ? 0 : 24)) * fb;
        WFX_TRY(wfx_dev_notch_fir_only(ctx, rin, kind, at_end ? SH_FMM_HR : SH_FMM_HR + 24, p.notch_b, p.notch_a, tr, at_end ? 1 : 0, use_ext ? ext18 : nullptr));
        WFX_HIP(ctx, hipMemcpyAsync(audio + SH_FMM_HF + n_own, tr + (at_end ? 0 : 24), SH_FMM_HF * 8, hipMemcpyDeviceToDevice, ctx->stream));
        WFX_TRY(wfx_fmm_shard_up(ctx, in, raw0, kind, p.notch_b, p.notch_a, use_ext ? ext18 : nullptr, audio, audio0, pl.n, pl.gb_lo, pl.gb_hi, ds));
        std::vector<wfx_xfer> xs;
        int rc = 0;
        fmm_weight_messages(pl, [&](int s, int r, int lev, long long box, long long cnt) {
            if (rc != 0 || (s != me && r != me)) return;
            double *ptr = nullptr;
            rc = wfx_fmm_shard_weights(ctx, pl.n, lev, box, &ptr);
            wfx_xfer x{};
            x.peer = s == me ? r : s;
            if (s == me) {
                x.send = ptr;
                x.send_bytes = (size_t)cnt * 256;
            } else {
                x.recv = ptr;
                x.recv_bytes = (size_t)cnt * 256;
            }
            xs.push_back(x);
        });
        WFX_TRY(rc);
        wfx_comm_label(c, "fmm weights");
        return wfx_comm_exchange(c, ctx, xs.data(), (int)xs.size());
    }
    if (ph == 1) {
        WFX_TRY(wfx_dev_select_sharded_ws(ctx, &sh->ws));
        WFX_TRY(wfx_fmm_shard_down(ctx, audio, audio0, pl.n, pl.gb_lo, pl.gb_hi, (double *)sh->b_env.p, (long long)pl.own_lo, sh->ws));
        // four envelope values across every seam between two ranks, either way (r's last four to r + 1, r + 1's first four to r)
        const int shf = pl.fg.ltop - pl.fg.lg;
        std::vector<wfx_xfer> xs;
        for (int r = 0; r + 1 < W; ++r) {
            if (me != r && me != r + 1) continue;
            long long a, b;
            fmm_rank_boxes(pl, r, &a, &b);
            const long long seam_wg = b << shf;                             // first workgroup of rank r + 1
            double *last4 = nullptr, *first4 = nullptr;
            WFX_TRY(wfx_fmm_shard_edges(ctx, pl.n, seam_wg - 1, &last4));
            WFX_TRY(wfx_fmm_shard_edges(ctx, pl.n, seam_wg, &first4));
            last4 += 4;
            wfx_xfer x{}, y{};
            x.peer = y.peer = me == r ? r + 1 : r;
            if (me == r) {
                x.send = last4;
                x.send_bytes = 32;
                y.recv = first4;
                y.recv_bytes = 32;
            } else {
                x.recv = last4;
                x.recv_bytes = 32;
                y.send = first4;
                y.send_bytes = 32;
            }
            xs.push_back(x);
            xs.push_back(y);
        }
        wfx_comm_label(c, "fmm seams");
        return wfx_comm_exchange(c, ctx, xs.data(), (int)xs.size());
    }
    // ph == 2: the medians at the seams, then the select's first all-reduce
    WFX_TRY(wfx_fmm_shard_seams(ctx, pl.n, pl.gb_lo, pl.gb_hi, (double *)sh->b_env.p, (long long)pl.own_lo, sh->ws));
    wfx_comm_label(c, "select level 0");
    return wfx_comm_allreduce_u32(c, ctx, sh->ws, WFX_SEL_BINS);
}

static int run_phase(wfx_shard *sh, int ph)
{
    wfx_ctx *ctx = sh->ctx;
    wfx_comm *c = sh->comm;
    shard_plan &pl = sh->pl;
    const wfx_decode_params &p = sh->dp;
    const int me = pl.rank, W = pl.world;
    const uint64_t n_own = pl.own_hi - pl.own_lo, n_seg = pl.seg_hi - pl.seg_lo;
    wfx_dev_scalars *ds = (wfx_dev_scalars *)ctx->b_scal.p;
    const void *in = sh->ext_in ? sh->ext_in : sh->b_in.p;
    if (pl.single) {
        // rank 0: the fused one-GPU decode of the whole capture on this context; then the scalars (levels, peaks, start frame) to
        // every rank, so that wfx_shard_result answers everywhere as it does after a distributed decode
        if (me == 0) {
            WFX_TRY(wfx_decode_attach(ctx, in, &sh->dp));
            WFX_TRY(wfx_decode_run(ctx));
            ds = (wfx_dev_scalars *)ctx->b_scal.p;
        }
        std::vector<wfx_xfer> xs;
        for (int r = 1; r < W; ++r) {
            wfx_xfer x{};
            x.peer = me == 0 ? r : 0;
            if (me == 0) {
                x.send = ds;
                x.send_bytes = sizeof(wfx_dev_scalars);
                xs.push_back(x);
            } else if (r == me) {
                x.recv = ds;
                x.recv_bytes = sizeof(wfx_dev_scalars);
                xs.push_back(x);
            }
        }
        return wfx_comm_exchange(c, ctx, xs.data(), (int)xs.size());
    }
    if (pl.cols) return run_phase_cols(sh, ph);
    if (pl.fmm) {
        if (pl.rs) {
            if (ph <= 1) return run_phase_rs(sh, ph);
            ph -= 2;
        }
        if (ph <= 2) return run_phase_fmm(sh, ph);
        if (pl.rs) ph += 4;                 // (the transposing plans' numbering below: 0 .. 3 are THEIR resampler's)
        ph += 2;                            // 3 .. 6 = the select's level 1, its candidates, finish + quantise + gather, rank 0's tail (cases 9 .. 12)
    }
    double *audio = (double *)sh->b_audio.p;
    double *env = (double *)sh->b_env.p;
    uint8_t *dig_own = (uint8_t *)sh->b_dig.p + (me == 0 ? pl.own_lo : 0);
    if (pl.padded && !sh->ghat_ready) {
        // ---- the padded convolution's kernel, transformed once per shard: rows -> E1 -> pass 1 -> E2 -> slab passes ----
        wfx_dist &dk = pl.split_kernel ? sh->dHk : sh->dH;
        const wfx_dist_geom &gk = pl.split_kernel ? pl.gk : pl.g;
        if (ph == 0) {
            if (pl.plain)
                WFX_TRY(wfx_dev_hilbert_kernel_rows_real(ctx, (cplx *)sh->b_grow.p, (long long)gk.rows[me] * pl.Ms, (long long)gk.nrows(me) * pl.Ms, (long long)pl.n, pl.Kp));
            else
                WFX_TRY(wfx_dev_hilbert_kernel_rows(ctx, (cplx *)sh->b_grow.p, (long long)gk.rows[me] * pl.Ms, (long long)gk.nrows(me) * pl.Ms, (long long)pl.n, pl.Kp));
            return dk.fwd_pack_exchange(c, sh->b_grow.p);
        }
        if (ph == 1) return dk.fwd_pass1_exchange(c, 0);
        if (ph == 2) {
            cplx *G = nullptr;
            WFX_TRY(dk.fwd_slab(0, &G));
            if (pl.plain)      // the real kernel's table: one double per slab entry, untangled from its packed transform
                WFX_TRY(wfx_dist_real_untangle(ctx, pl.g, G, pl.Kp, (double *)sh->b_ghat.p));
            else
                WFX_HIP(ctx, hipMemcpyAsync(sh->b_ghat.p, G, (size_t)sh->dH.slab_points() * sizeof(cplx), hipMemcpyDeviceToDevice, ctx->stream));
            // the kernel's transform exists from here on: a decode that aborts in a later phase and is retried must NOT re-run
            // these three phases (the forward half is re-bound to the audio rows below)
            if (pl.split_kernel) {       // the kernel's transform object and its rows have done their work
                WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
                sh->dHk.release();
                free_buf(sh->b_grow);
                sh->ghat_pending = true;
                return 0;
            }
            // from here on the forward half reads the audio rows (re-binding rebuilds the E1 lists: a host synchronisation, once)
            WFX_TRY(sh->dH.bind(audio + (pl.own_lo - pl.seg_lo), (cplx *)sh->b_v.p, sh->dH.fwd_result_index()));
            sh->ghat_pending = true;
            return 0;
        }
        ph -= 3;
    }
    if (!pl.resample) ph += 4;              // phases 0..3 are the resampler's
    if (pl.padded && ph >= 8) {             // one phase more than the unpadded form: 8 = unpack + wrap, 9.. = the old 8..
        if (ph == 8) {
            // the cyclic result's V[K] is V[0] (H[N - 1] sits in its .y): it lives on rank 0, the rank that holds the capture's
            // last pair needs it where its rows continue into the padding (the padded convolution left garbage there)
            WFX_TRY(sh->dH.inv_unpack((cplx *)sh->b_v.p));
            cplx *vown = (cplx *)sh->b_v.p + SH_VHALO;
            wfx_xfer x{};
            if (pl.plain) return wfx_comm_exchange(c, ctx, &x, 0);         // (odd length: a LINEAR convolution with both signs of the lag in the kernel -- nothing wraps)
            x.peer = me == 0 ? pl.wrap_rank : 0;
            if (me == 0) {
                x.send = vown;
                x.send_bytes = sizeof(cplx);
            }
            if (me == pl.wrap_rank) {
                x.recv = vown + (pl.K - (long long)pl.g.rows[me] * pl.Ms);
                x.recv_bytes = sizeof(cplx);
            }
            if (me == 0 && pl.wrap_rank == 0) x.peer = 0;
            return (x.send_bytes || x.recv_bytes) ? wfx_comm_exchange(c, ctx, &x, 1) : wfx_comm_exchange(c, ctx, &x, 0);
        }
        ph -= 1;
    }
    switch (ph) {
    case 0: {   // a4 + a5 first exchange: input rows -> columns
        WFX_HIP(ctx, hipMemsetAsync(ds, 0, sizeof(wfx_dev_scalars), ctx->stream));
        if (pl.in_kind == WFX_IN_I16_STEREO) WFX_TRY(wfx_dev_merge(ctx, (const int16_t *)in, pl.in_hi - pl.in_lo, (double *)sh->b_merged.p));
        return sh->dF.fwd_pack_exchange(c, pl.in_kind == WFX_IN_I16_STEREO ? sh->b_merged.p : in);
    }
    case 1: return sh->dF.fwd_pass1_exchange(c, pl.in_kind == WFX_IN_I16_MONO ? 2 : 0);
    case 2: {   // spectrum -> scipy.signal.resample's bin copy -> inverse transform's slab passes
        cplx *Z = nullptr;
        // down-sampling reads the bins [0, n/2] and their mirrors only: the last forward pass does not store the rest
        const long long nmin = (long long)(pl.n0 < pl.n ? pl.n0 : pl.n), half = nmin / 2;
        WFX_TRY(sh->dF.fwd_slab(0, &Z, half, pl.M1 - half));
        WFX_TRY(wfx_dist_resample_glue(ctx, pl.g, Z, (long long)pl.n0, (long long)pl.n, sh->dI.slab_buffer(0)));
        return sh->dI.inv_slab_exchange(c, sh->dI.slab_buffer(0));
    }
    case 3: return sh->dI.inv_pass1_exchange(c, (cplx *)sh->b_res.p);
    case 4: {   // a6 notch on [seg_lo, seg_hi), then the Hilbert transform's first exchange
        const void *nin = in;
        int nkind = pl.in_kind;
        if (pl.resample) {
            WFX_TRY(sh->dI.inv_unpack((cplx *)sh->b_res.p));
            // b_res holds samples [own_lo - 32, own_hi + 32) (circular); the segment starts at seg_lo
            nin = (const double *)sh->b_res.p + (pl.seg_lo + SH_HALO - pl.own_lo);
            nkind = WFX_IN_F64_MONO;
        } else {
            WFX_HIP(ctx, hipMemsetAsync(ds, 0, sizeof(wfx_dev_scalars), ctx->stream));
            if (pl.in_kind == WFX_IN_I16_STEREO && n_seg) {
                WFX_TRY(wfx_dev_merge(ctx, (const int16_t *)in, n_seg, (double *)sh->b_merged.p));
                nin = sh->b_merged.p;
                nkind = WFX_IN_F64_MONO;
            }
        }
        const int flags = (pl.seg_lo == 0 ? 1 : 0) | (pl.seg_hi == pl.n ? 2 : 0);
        double ext18[18];
        const bool use_ext = p.has_ext && !pl.resample && pl.in_kind != WFX_IN_I16_STEREO;
        for (int i = 0; i < 9; ++i) {
            ext18[i] = p.ext_left[i];
            ext18[9 + i] = p.ext_right[i];
        }
        if (n_seg) WFX_TRY(wfx_dev_notch_fir_only(ctx, nin, nkind, n_seg, p.notch_b, p.notch_a, audio, flags, use_ext ? ext18 : nullptr));
        return sh->dH.fwd_pack_exchange(c, audio + (pl.own_lo - pl.seg_lo));
    }
    case 5: return sh->dH.fwd_pass1_exchange(c, 1);
    case 6: {
        cplx *G = nullptr;
        if (pl.plain) {     // real samples, real kernel: the glue between the packed forward and inverse transforms, in place
            WFX_TRY(sh->dH.fwd_slab(0, &G));
            WFX_TRY(wfx_dist_real_conv_glue(ctx, pl.g, G, pl.Kp, (const double *)sh->b_ghat.p));
        } else if (pl.padded)
            WFX_TRY(sh->dH.fwd_slab(0, &G, 0, 0, (const cplx *)sh->b_ghat.p));
        else
            WFX_TRY(sh->dH.fwd_slab(1, &G));
        return sh->dH.inv_slab_exchange(c, G);
    }
    case 7: return sh->dH.inv_pass1_exchange(c, (cplx *)sh->b_v.p);
    case 8: {   // a7 envelope + median, level-0 histogram; first all-reduce
        if (!pl.padded) WFX_TRY(sh->dH.inv_unpack((cplx *)sh->b_v.p));         // (padded: done with the wrap, one phase earlier)
        WFX_TRY(wfx_dev_select_sharded_ws(ctx, &sh->ws));
        // pointers indexed by global pair / sample index (the rank's rows start at pair rows[me] Ms = own_lo / 2 when it owns samples)
        const cplx *Vg = (const cplx *)sh->b_v.p - ((long long)(pl.own_lo / pl.spp) - SH_VHALO);
        const double *xg = audio - (long long)pl.seg_lo;
        if (n_own && pl.plain) WFX_TRY(wfx_dev_env_median_block_plain(ctx, Vg, xg, pl.n, pl.own_lo, pl.own_hi, env, sh->ws));
        if (n_own && !pl.plain) WFX_TRY(wfx_dev_env_median_block(ctx, Vg, xg, pl.n, pl.own_lo, pl.own_hi, env, sh->ws));
        wfx_comm_label(c, "select level 0");
        return wfx_comm_allreduce_u32(c, ctx, sh->ws, WFX_SEL_BINS);
    }
    case 9: {
        const uint64_t ranks[4] = {p.rank_lo[0], p.rank_lo[1], p.rank_hi[0], p.rank_hi[1]};
        WFX_TRY(wfx_dev_select_l1(ctx, env, n_own, ranks, sh->ws, ds));
        wfx_comm_label(c, "select level 1");
        return wfx_comm_allreduce_u32(c, ctx, sh->ws + WFX_SEL_H1_OFFSET, WFX_SEL_H1_WORDS);
    }
    case 10: {
        WFX_TRY(wfx_dev_select_compact_block(ctx, env, n_own, sh->ws, ds, sh->b_blk.p, sh->cap));
        wfx_comm_label(c, "select candidates");
        return wfx_comm_allgather(c, ctx, sh->b_blk.p, sh->b_blks.p, wfx_select_block_bytes(sh->cap));
    }
    case 11: {  // a8 finish + quantise; the one gather of the stream
        WFX_TRY(wfx_dev_select_finish_blocks(ctx, sh->ws, ds, sh->b_blks.p, W, sh->cap, p.gamma_lo, p.gamma_hi, (unsigned *)sh->b_flags.p));
        if (n_own) WFX_TRY(wfx_dev_quantise(ctx, env, n_own, ds, dig_own, ds));
        std::vector<wfx_xfer> xs;
        if (me == 0) {
            for (int s = 1; s < W; ++s) {
                uint64_t lo, hi;
                if (pl.fmm) {
                    fmm_rank_range(pl, s, &lo, &hi);
                } else {
                    lo = (uint64_t)pl.spp * (uint64_t)pl.g.rows[s] * (uint64_t)pl.Ms;
                    hi = (uint64_t)pl.spp * (uint64_t)pl.g.rows[s + 1] * (uint64_t)pl.Ms;
                }
                lo = lo < pl.n ? lo : pl.n;                      // (padded form: rows beyond the capture hold no samples)
                hi = hi < pl.n ? hi : pl.n;
                wfx_xfer a{};
                a.peer = s;
                a.recv = (uint8_t *)sh->b_dig.p + lo;
                a.recv_bytes = hi - lo;
                if (a.recv_bytes) xs.push_back(a);
                wfx_xfer b{};
                b.peer = s;
                b.recv = (unsigned long long *)sh->b_nan.p + s;
                b.recv_bytes = 8;
                xs.push_back(b);
            }
        } else {
            wfx_xfer a{};
            a.peer = 0;
            a.send = dig_own;
            a.send_bytes = n_own;
            if (n_own) xs.push_back(a);
            wfx_xfer b{};
            b.peer = 0;
            b.send = &ds->nan_count;
            b.send_bytes = 8;
            xs.push_back(b);
        }
        wfx_comm_label(c, "stream gather");
        return wfx_comm_exchange(c, ctx, xs.data(), (int)xs.size());
    }
    case 12: {  // a9 + a10 on rank 0
        if (me != 0) return 0;
        if (W > 1) WFX_TRY(wfx_dev_add_u64(ctx, &ds->nan_count, (const unsigned long long *)sh->b_nan.p + 1, W - 1));
        const int w = p.width;
        const int h_max = (int)(pl.n / (uint64_t)w);
        WFX_TRY(wfx_reserve(ctx, ctx->b_img, (size_t)w * 4 * (size_t)(h_max > 0 ? h_max : 1)));
        WFX_TRY(wfx_dev_sync_pick(ctx, (const uint8_t *)sh->b_dig.p, pl.n, p.n1, p.n0_gap, p.mindistance, p.frame_samples, w, ds));
        return wfx_dev_image(ctx, (const uint8_t *)sh->b_dig.p, pl.n, w, h_max, ds, (uint8_t *)ctx->b_img.p, ctx->h_scal);
    }
    default: return wfx_fail(ctx, WFX_ERR_BAD_ARG, "no such phase");
    }
}

// ---- plan check without a GPU -----------------------------------------------------------------------------
// Builds every rank's exchange lists and copy descriptors of one distributed transform with fake buffer addresses and checks
// what a real run relies on: the k-th message rank a sends to rank b has the size of the k-th message b expects from a; what a
// rank receives lands inside its buffers, without overlaps, and adds up to the layout's size; every copy stays inside its
// source and destination.  (The in-process communicator makes the first check at run time; this one also covers sizes no
// test GPU holds, e.g. the 60-minute captures at 8 ranks.)
struct dry_region {
    unsigned long long lo, hi;
};

static bool dry_inside(const std::vector<dry_region> &regs, unsigned long long lo, unsigned long long hi)
{
    for (const dry_region &r : regs)
        if (lo >= r.lo && hi <= r.hi) return true;
    return false;
}

// every rank's transform object of one distributed transform, planned with fake addresses (no device memory)
static int dry_build(const shard_plan &pl0, long long L, int es, int hb, int ha, bool fwd, bool inv, bool cols, int rows_used, std::vector<wfx_dist> &d,
                     std::vector<std::vector<dry_region>> &regs)
{
    const int W = pl0.world;
    d.clear();
    d.resize(W);
    regs.assign(W, {});
    int rc = 0;
    for (int r = 0; r < W && rc == 0; ++r) {
        wfx_dist_geom g;
        if (!wfx_dist_make_geom(g, W, r, pl0.g.ra1, pl0.g.rb1, rows_used)) return wfx_fail(nullptr, WFX_ERR_BAD_ARG, "dry run: geometry");
        const unsigned long long base = (unsigned long long)(r + 1) << 44;
        rc = d[r].init(nullptr, g, L, es, hb, ha, true, base, cols ? pl0.nchunk : 1);
        if (rc) break;
        std::vector<std::pair<unsigned long long, unsigned long long>> b;
        d[r].buffers(b);
        for (auto &x : b) regs[r].push_back({x.first, x.first + x.second});
        const unsigned long long rin = base + (6ull << 36), rout = base + (7ull << 36);
        if (cols) {
            const long long in_rs = d[r].w + 8, out_rs = (long long)hb + d[r].w + ha + 2;      // (any strides that hold a row)
            regs[r].push_back({rin, rin + (unsigned long long)pl0.g.R1 * in_rs * es});
            regs[r].push_back({rout, rout + (unsigned long long)pl0.g.R1 * out_rs * 16});
            rc = d[r].bind_cols(fwd ? (const void *)rin : nullptr, in_rs, inv ? (cplx *)rout : nullptr, out_rs, d[r].fwd_result_index());
        } else {
            regs[r].push_back({rin, rin + (unsigned long long)d[r].nr * d[r].M * es});
            regs[r].push_back({rout, rout + (unsigned long long)(hb + (long long)d[r].nr * d[r].M + ha) * 16});
            rc = d[r].bind(fwd ? (const void *)rin : nullptr, inv ? (cplx *)rout : nullptr, d[r].fwd_result_index());
        }
    }
    return rc;
}

static int dry_check_transform(const shard_plan &pl0, const wfx_decode_params *p, long long L, int es, int hb, int ha, bool fwd, bool inv, const char *name,
                               int rows_used = 0, bool cols = false)
{
    const int W = pl0.world;
    std::vector<wfx_dist> d;
    std::vector<std::vector<dry_region>> regs;
    int rc = dry_build(pl0, L, es, hb, ha, fwd, inv, cols, rows_used, d, regs);
    for (int e = 1; e <= 5 && rc == 0; ++e) {
        if ((e <= 2 && !fwd) || (e >= 3 && !inv)) continue;
        if (cols ? (e == 1 || e == 4) : e == 5) continue;         // columns layout: no E1 / E4, a halo exchange (5) instead
        const int pes = e == 1 ? es : 16;
        for (int src = 0; src < W && rc == 0; ++src)
            for (int dst = 0; dst < W && rc == 0; ++dst) {
                std::vector<size_t> snd, rcv;
                for (const wfx_xfer &x : d[src].xfers(e))
                    if (x.peer == dst && x.send_bytes) snd.push_back(x.send_bytes);
                for (const wfx_xfer &x : d[dst].xfers(e))
                    if (x.peer == src && x.recv_bytes) rcv.push_back(x.recv_bytes);
                if (snd != rcv)
                    rc = wfx_fail(nullptr, WFX_ERR_COMM, "dry run (%s, exchange %d, world %d): rank %d -> rank %d message sizes do not match", name, e, W, src, dst);
            }
        for (int r = 0; r < W && rc == 0; ++r) {
            std::vector<dry_region> got;
            unsigned long long total = 0;
            for (const wfx_xfer &x : d[r].xfers(e)) {
                if (x.send_bytes && !dry_inside(regs[r], (unsigned long long)x.send, (unsigned long long)x.send + x.send_bytes))
                    rc = wfx_fail(nullptr, WFX_ERR_COMM, "dry run (%s, exchange %d): rank %d sends from outside its buffers", name, e, r);
                if (!x.recv_bytes) continue;
                const unsigned long long lo = (unsigned long long)x.recv, hi = lo + x.recv_bytes;
                if (!dry_inside(regs[r], lo, hi)) rc = wfx_fail(nullptr, WFX_ERR_COMM, "dry run (%s, exchange %d): rank %d receives outside its buffers", name, e, r);
                for (const dry_region &o : got)
                    if (lo < o.hi && o.lo < hi) rc = wfx_fail(nullptr, WFX_ERR_COMM, "dry run (%s, exchange %d): rank %d receives twice into the same bytes", name, e, r);
                got.push_back({lo, hi});
                total += x.recv_bytes;
            }
            const unsigned long long R1 = pl0.g.R1, Rd = (rows_used > 0 && rows_used < pl0.g.R1) ? (unsigned long long)rows_used : R1;     // rows that travel in E1
            // (one rank: the first pass reads the rows in place and the last one writes them in place -- E1 and E4 carry the halo only)
            const unsigned long long want = e == 1 ? (W == 1 ? 0ull : Rd * d[r].w * (unsigned long long)es) : e == 2 ? (unsigned long long)d[r].M * d[r].B * 16
                                          : e == 3 ? R1 * d[r].w * 16ull : e == 5 ? R1 * (unsigned long long)(hb + ha) * 16ull
                                          : ((W == 1 ? 0ull : (unsigned long long)d[r].nr * d[r].M) + hb + ha) * 16ull;
            if (rc == 0 && total != want)
                rc = wfx_fail(nullptr, WFX_ERR_COMM, "dry run (%s, exchange %d): rank %d receives %llu bytes, its layout holds %llu", name, e, r, total, want);
            if (e == 2 && rc == 0) {
                // the first pass's scatter map: every output k1 has a target, each target column lies inside a buffer, no two coincide
                std::vector<unsigned long long> bases;
                for (const mr_qmap &m : d[r].first_pass_map()) {
                    if (!m.base || m.stride <= 0 || !dry_inside(regs[r], m.base, m.base + ((unsigned long long)(d[r].w - 1) * m.stride + 1) * 16))
                        rc = wfx_fail(nullptr, WFX_ERR_COMM, "dry run (%s): rank %d's first pass stores outside its buffers", name, r);
                    bases.push_back(m.base);
                }
                std::sort(bases.begin(), bases.end());
                if (std::adjacent_find(bases.begin(), bases.end()) != bases.end() || (int)bases.size() != pl0.g.R1)
                    rc = wfx_fail(nullptr, WFX_ERR_COMM, "dry run (%s): rank %d's first pass stores two outputs to one place", name, r);
            }
            for (const wfx_dist_piece &q : d[r].pieces(e)) {
                if (rc) break;
                const bool kmap = e == 2 || e == 3;
                // extents: copy2d rows x cols with row strides; gather / scatter by k1 touch the whole [rows][R1] / [R1][rows] array
                unsigned long long slo = q.src, shi, dlo = q.dst, dhi;
                if (!kmap) {
                    shi = slo + ((unsigned long long)(q.rows - 1) * q.src_rs + q.cols) * pes;
                    dhi = dlo + ((unsigned long long)(q.rows - 1) * q.dst_rs + q.cols) * pes;
                } else if (e == 2) {
                    shi = slo + (unsigned long long)q.rows * q.src_rs * 16;
                    dhi = dlo + (unsigned long long)q.rows * q.B * 16;
                } else {
                    shi = slo + (unsigned long long)q.rows * q.B * 16;
                    dhi = dlo + (unsigned long long)R1 * q.dst_rs * 16;
                }
                if (q.rows > 0 && q.cols > 0 && (!dry_inside(regs[r], slo, shi) || !dry_inside(regs[r], dlo, dhi)))
                    rc = wfx_fail(nullptr, WFX_ERR_COMM, "dry run (%s, exchange %d): a copy of rank %d leaves its buffers", name, e, r);
            }
        }
    }
    for (int r = 0; r < W; ++r) d[r].release();
    (void)p;
    return rc;
}

extern "C" int wfx_shard_dry_run(const wfx_decode_params *p, int world)
{
    shard_plan pl;
    WFX_TRY(make_plan(nullptr, p, world, 0, pl));
    // every rank's own layout must tile the capture
    uint64_t next = 0, next_in = 0;
    for (int r = 0; r < world; ++r) {
        shard_plan q;
        WFX_TRY(make_plan(nullptr, p, world, r, q));
        if (q.cols) {
            // columns layout: the ranks' column ranges tile [0, Ms) (and [0, M1s)); a rank's samples are those columns of every row
            if (2ull * (uint64_t)q.cH0 != next) return wfx_fail(nullptr, WFX_ERR_COMM, "dry run: rank %d's columns start at sample %llu of a row, expected %llu", r, 2ull * (unsigned long long)q.cH0, (unsigned long long)next);
            next = 2ull * (uint64_t)(q.cH0 + q.wH);
            if (q.resample) {
                if (2ull * (uint64_t)q.cF0 != next_in) return wfx_fail(nullptr, WFX_ERR_COMM, "dry run: rank %d's input columns do not follow rank %d's", r, r - 1);
                next_in = 2ull * (uint64_t)(q.cF0 + q.wF);
            }
            continue;
        }
        if (q.own_lo != next) return wfx_fail(nullptr, WFX_ERR_COMM, "dry run: rank %d starts at sample %llu, expected %llu", r, (unsigned long long)q.own_lo, (unsigned long long)next);
        next = q.own_hi;
        if (q.resample) {
            if (q.in_lo != next_in) return wfx_fail(nullptr, WFX_ERR_COMM, "dry run: rank %d's input starts at frame %llu, expected %llu", r, (unsigned long long)q.in_lo, (unsigned long long)next_in);
            next_in = q.in_hi;
        }
    }
    if (pl.cols) {
        if (next != 2ull * (uint64_t)pl.Ms || (pl.resample && next_in != 2ull * (uint64_t)pl.M1s) ||
            (pl.padded ? (uint64_t)pl.g.R1 * 2ull * (uint64_t)pl.Ms < p->n : (uint64_t)pl.g.R1 * 2ull * (uint64_t)pl.Ms != p->n))
            return wfx_fail(nullptr, WFX_ERR_COMM, "dry run: the ranks' columns do not cover the rows");
    } else if (next != p->n || (pl.resample && next_in != p->n0))
        return wfx_fail(nullptr, WFX_ERR_COMM, "dry run: the ranks' ranges do not cover the capture");
    if (pl.single) return 0;              // rank 0 alone: one message of scalars per peer, nothing to cross-check
    if (pl.fmm) {
        // plan 3: every rank walks ONE enumeration of (sender, receiver, level, box): the two ends of a message agree by construction; what is
        // checked here is that every box a rank's kernels read beyond its own range is delivered exactly once and by its owner
        for (int tree = 0; tree < (pl.rs ? 2 : 1); ++tree) {
            const int tlg = tree ? pl.fgr.lg : pl.fg.lg, tL = tree ? pl.fgr.L : pl.fg.L;
            for (int r = 0; r < world; ++r) {
                long long a, b;
                fmm_rank_boxes_lv(pl, tlg, r, &a, &b);
                for (int lev = tlg + 1; lev <= tL; ++lev) {
                    const long long nbl = 1ll << lev, lo = a << (lev - tlg), hi = b << (lev - tlg);
                    std::vector<long long> need, got;
                    for (int j = 0; j < 3; ++j) {
                        for (long long raw : {lo - 3 + j, hi + j}) {
                            const long long box = (raw + nbl) & (nbl - 1);
                            if (box < lo || box >= hi) need.push_back(box);
                        }
                    }
                    fmm_weight_messages_lv(pl, tlg, tL, [&](int s, int rr, int lv, long long box, long long cnt) {
                        if (rr == r && lv == lev) {
                            long long sa, sb;
                            fmm_rank_boxes_lv(pl, tlg, s, &sa, &sb);
                            for (long long q = 0; q < cnt; ++q)
                                if (((box + q) >> (lev - tlg)) >= sa && ((box + q) >> (lev - tlg)) < sb && box + q < nbl) got.push_back(box + q);
                        }
                    });
                    std::sort(need.begin(), need.end());
                    std::sort(got.begin(), got.end());
                    if (need != got)
                        return wfx_fail(nullptr, WFX_ERR_COMM, "dry run (multipole plan, %s tree): rank %d, level %d: the boxes delivered are not the boxes read",
                                        tree ? "resampler" : "Hilbert", r, lev);
                }
            }
        }
        return 0;
    }
    if (pl.cols) {
        if (pl.resample) {
            WFX_TRY(dry_check_transform(pl, p, pl.M1, pl.in_kind == WFX_IN_I16_MONO ? 4 : 16, 0, 0, true, false, "resample forward (columns)", 0, true));
            WFX_TRY(dry_check_transform(pl, p, pl.K, 16, pl.hs / 2, pl.hs / 2, false, true, "resample inverse (columns)", 0, true));
        }
        return dry_check_transform(pl, p, pl.Kp, 16, SH_VHALO, SH_VHALO, true, true, "hilbert (columns)", 0, true);
    }
    if (pl.resample) {
        WFX_TRY(dry_check_transform(pl, p, pl.M1, pl.in_kind == WFX_IN_I16_MONO ? 4 : 16, 0, 0, true, false, "resample forward"));
        WFX_TRY(dry_check_transform(pl, p, pl.K, 16, SH_HALO / 2, SH_HALO / 2, false, true, "resample inverse"));
    }
    if (pl.split_kernel) WFX_TRY(dry_check_transform(pl, p, pl.Kp, 16, 0, 0, true, false, "hilbert kernel (all rows)"));
    return dry_check_transform(pl, p, pl.Kp, 16, SH_VHALO, SH_VHALO, true, true, pl.plain ? "hilbert (odd length)" : (pl.padded ? "hilbert (padded)" : "hilbert"),
                               pl.g.rows_used);
}

// ---- what the plan puts on the wire (host only) -----------------------------------------------------------------
extern "C" int wfx_shard_wire_plan(const wfx_decode_params *p, int world, wfx_wire_entry *out, int cap)
{
    shard_plan pl;
    WFX_TRY(make_plan(nullptr, p, world, 0, pl));
    const int W = world;
    int n = 0;
    auto put = [&](const char *name, unsigned long long total, unsigned long long max_rank, unsigned long long max_link) {
        if (out && n < cap) {
            memset(&out[n], 0, sizeof out[n]);
            snprintf(out[n].name, sizeof out[n].name, "%s", name);
            out[n].total_bytes = total;
            out[n].max_rank_bytes = max_rank;
            out[n].max_link_bytes = max_link;
        }
        ++n;
    };
    if (pl.single) {
        put("scalars", (unsigned long long)(W - 1) * sizeof(wfx_dev_scalars), (unsigned long long)(W - 1) * sizeof(wfx_dev_scalars), sizeof(wfx_dev_scalars));
        return n;
    }
    if (pl.fmm) {
        if (pl.rs) {
            std::vector<unsigned long long> sent((size_t)W, 0ull), link((size_t)W * W, 0ull);
            fmm_weight_messages_lv(pl, pl.fgr.lg, pl.fgr.L, [&](int s, int r, int, long long, long long cnt) {
                sent[(size_t)s] += (unsigned long long)cnt * 256;
                link[(size_t)s * W + r] += (unsigned long long)cnt * 256;
            });
            for (int s2 = 0; s2 < W; ++s2)
                for (int r = 0; r < W; ++r) {
                    if (s2 == r) continue;
                    long long a, b;
                    fmm_rank_boxes_lv(pl, pl.fgr.lg, s2, &a, &b);
                    sent[(size_t)s2] += (unsigned long long)(b - a) * 8;
                    link[(size_t)s2 * W + r] += (unsigned long long)(b - a) * 8;
                }
            unsigned long long total = 0, mr = 0, ml = 0;
            for (int r = 0; r < W; ++r) {
                total += sent[(size_t)r];
                mr = std::max(mr, sent[(size_t)r]);
            }
            for (unsigned long long v : link) ml = std::max(ml, v);
            put("resampler weights", total, mr, ml);
            const unsigned long long hb = SH_FMM_HR * 8ull;
            put("resampled halos", W > 1 ? 2 * hb * W : 0ull, W > 1 ? 2 * hb : 0ull, W > 1 ? (W == 2 ? 2 * hb : hb) : 0ull);
        }
        std::vector<unsigned long long> sent((size_t)W, 0ull), link((size_t)W * W, 0ull);
        fmm_weight_messages(pl, [&](int s, int r, int, long long, long long cnt) {
            sent[(size_t)s] += (unsigned long long)cnt * 256;
            link[(size_t)s * W + r] += (unsigned long long)cnt * 256;
        });
        unsigned long long total = 0, mr = 0, ml = 0;
        for (int r = 0; r < W; ++r) {
            total += sent[(size_t)r];
            mr = std::max(mr, sent[(size_t)r]);
        }
        for (unsigned long long v : link) ml = std::max(ml, v);
        put("fmm weights", total, mr, ml);
        put("fmm seams", W > 1 ? 64ull * (W - 1) : 0ull, W > 1 ? 64ull : 0ull, W > 1 ? 32ull : 0ull);
        const unsigned long long ar0 = 2ull * (W - 1) * WFX_SEL_BINS * 4 / W, ar1 = 2ull * (W - 1) * WFX_SEL_H1_WORDS * 4 / W;
        put("select level 0", ar0 * W, ar0, WFX_SEL_BINS * 4ull / W);
        put("select level 1", ar1 * W, ar1, WFX_SEL_H1_WORDS * 4ull / W);
        uint64_t cap_c = SH_CAND_CAP / 4;
        {
            const uint64_t want_c = p->n / (512ull * (uint64_t)W);
            while (cap_c < want_c && cap_c < (1ull << 20)) cap_c *= 2;
        }
        const unsigned long long blk = wfx_select_block_bytes(cap_c);
        put("select candidates", blk * W * (W - 1), blk * (W - 1), blk);
        uint64_t lo0, hi0;
        fmm_rank_range(pl, 0, &lo0, &hi0);
        unsigned long long big = 0;
        for (int r = 1; r < W; ++r) {
            uint64_t lo, hi;
            fmm_rank_range(pl, r, &lo, &hi);
            big = std::max(big, (unsigned long long)(hi - lo) + 8);
        }
        put("stream gather", W > 1 ? (unsigned long long)(p->n - (hi0 - lo0)) + 8ull * (W - 1) : 0ull, big, big);
        return n;
    }
    // one transform's exchanges, from every rank's (dry) exchange lists
    auto transform = [&](const char *tag, long long L, int es, int hb, int ha, bool fwd, bool inv, int rows_used) -> int {
        std::vector<wfx_dist> d;
        std::vector<std::vector<dry_region>> regs;
        WFX_TRY(dry_build(pl, L, es, hb, ha, fwd, inv, pl.cols, rows_used, d, regs));
        for (int e = 1; e <= 5; ++e) {
            if ((e <= 2 && !fwd) || (e >= 3 && !inv)) continue;
            if (pl.cols ? (e == 1 || e == 4) : e == 5) continue;
            // (order within a transform: E1, E2, E3, E4 | E2 and E3 of every k1 subset, halo)
            const int nsub = (e == 2 || e == 3) ? d[0].chunks() : 1;
            for (int c = 0; c < nsub; ++c) {
                unsigned long long total = 0, mr = 0, ml = 0;
                for (int r = 0; r < W; ++r) {
                    std::vector<unsigned long long> link((size_t)W, 0ull);
                    unsigned long long mine = 0;
                    for (const wfx_xfer &x : ((e == 2 || e == 3) ? d[r].xfers_chunk(e, c) : d[r].xfers(e)))
                        if (x.peer != r) {
                            mine += x.send_bytes;
                            link[(size_t)x.peer] += x.send_bytes;
                        }
                    total += mine;
                    mr = std::max(mr, mine);
                    for (unsigned long long v : link) ml = std::max(ml, v);
                }
                char name[24];
                snprintf(name, sizeof name, "%.15s %s", tag, e == 5 ? "halo" : (e == 1 ? "E1" : e == 2 ? "E2" : e == 3 ? "E3" : "E4"));
                put(name, total, mr, ml);
            }
        }
        for (int r = 0; r < W; ++r) d[r].release();
        return 0;
    };
    if (pl.resample) {
        WFX_TRY(transform("resample fwd", pl.M1, pl.in_kind == WFX_IN_I16_MONO ? 4 : 16, 0, 0, true, false, 0));
        const int h = pl.cols ? pl.hs / 2 : SH_HALO / 2;
        WFX_TRY(transform("resample inv", pl.K, 16, h, h, false, true, 0));
    }
    WFX_TRY(transform("hilbert", pl.Kp, 16, SH_VHALO, SH_VHALO, true, true, pl.g.rows_used));
    const unsigned long long ar0 = 2ull * (W - 1) * WFX_SEL_BINS * 4 / W, ar1 = 2ull * (W - 1) * WFX_SEL_H1_WORDS * 4 / W;
    put("select level 0", ar0 * W, ar0, WFX_SEL_BINS * 4ull / W);
    put("select level 1", ar1 * W, ar1, WFX_SEL_H1_WORDS * 4ull / W);
    {
        uint64_t want = p->n / (512ull * (uint64_t)W), capk = SH_CAND_CAP;
        while (capk < want && capk < (1ull << 20)) capk *= 2;
        const unsigned long long blk = wfx_select_block_bytes(capk);
        put("select candidates", (unsigned long long)W * (W - 1) * blk, (unsigned long long)(W - 1) * blk, blk);
    }
    {
        unsigned long long total = 0, big = 0;
        for (int r = 1; r < W; ++r) {
            shard_plan q;
            WFX_TRY(make_plan(nullptr, p, world, r, q));
            const unsigned long long nb = (q.cols ? cols_own(q) : q.own_hi - q.own_lo) + 8;
            total += nb;
            big = std::max(big, nb);
        }
        put("stream gather", total, big, big);
    }
    return n;
}

#define CHECK_SH(sh)                                                                                              \
    do {                                                                                                          \
        if (!(sh)) return wfx_fail(nullptr, WFX_ERR_BAD_ARG, "null shard");                                       \
        if (!wfx_ctx_alive((sh)->ctx)) return wfx_fail(nullptr, WFX_ERR_STATE, "the shard's context was destroyed"); \
        (void)hipSetDevice((sh)->ctx->device);                                                                    \
    } while (0)

extern "C" {

int wfx_shard_layout_query(const wfx_decode_params *p, int world, int rank, wfx_shard_layout *out)
{
    if (!out) return wfx_fail(nullptr, WFX_ERR_BAD_ARG, "null argument");
    shard_plan pl;
    WFX_TRY(make_plan(nullptr, p, world, rank, pl));
    memset(out, 0, sizeof *out);
    out->world = world;
    out->rank = rank;
    out->first_radix[0] = pl.g.ra1;
    out->first_radix[1] = pl.g.rb1;
    out->in_lo = pl.in_lo;
    out->in_hi = pl.in_hi;
    out->own_lo = pl.own_lo;
    out->own_hi = pl.own_hi;
    out->nseg = 1;
    out->plan = pl.single ? 0 : (pl.fmm ? 3 : (pl.cols ? 2 : 1));
    if (pl.fmm) out->in_halo = (unsigned)pl.in_halo;                              // frames beyond [in_lo, in_hi) on either side, ROUND THE CIRCLE (the capture's other end)
    out->plan_forced = pl.forced;
    out->model_single_s = pl.model_single;
    out->model_dist_compute_s = pl.model_comp;
    out->model_dist_wire_s = pl.model_wire;
    out->model_wire_bytes = pl.model_bytes;
    if (pl.fmm)
        snprintf(out->plan_reason, sizeof out->plan_reason, pl.rs ? "chunk-local multipole forms (resampler + Hilbert transform): ranks own boxes of level %d of a %d-level tree"
                                                                   : "chunk-local multipole form: ranks own boxes of level %d of a %d-level tree", pl.lgc, pl.fg.L);
    else if (pl.cols && !pl.single)
        snprintf(out->plan_reason, sizeof out->plan_reason, "columns layout, every transpose in %d k1 subset%s", pl.nchunk, pl.nchunk == 1 ? "" : "s");
    else
        snprintf(out->plan_reason, sizeof out->plan_reason, "%s", pl.single ? pl.single_reason : (pl.padded ? "rows layout (padded form)" : "rows layout"));
    if (pl.cols && world > 1) {
        out->nseg = pl.g.R1;
        out->own_seg_len = 2ull * (uint64_t)pl.wH;
        out->own_seg_stride = 2ull * (uint64_t)pl.Ms;
        out->in_halo = pl.resample ? 0 : pl.hs;
        out->in_seg_len = pl.resample ? 2ull * (uint64_t)pl.wF : 2ull * (uint64_t)pl.wH;
        out->in_seg_stride = pl.resample ? 2ull * (uint64_t)pl.M1s : 2ull * (uint64_t)pl.Ms;
    }
    return 0;
}

int wfx_shard_create(wfx_ctx *ctx, wfx_comm *comm, const wfx_decode_params *p, wfx_shard **out)
{
    if (!ctx || !comm || !p || !out) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null argument");
    *out = nullptr;
    (void)hipSetDevice(ctx->device);
    wfx_shard *sh = new wfx_shard();
    sh->ctx = ctx;
    sh->comm = comm;
    sh->dp = *p;
    int rc = make_plan(ctx, p, wfx_comm_world(comm), wfx_comm_rank(comm), sh->pl);
    const shard_plan &pl = sh->pl;
    {   // The select's candidates are the keys that share 22 leading bits with a percentile: ~ n / 8192 of them per octave of
        // envelope range in all, a world-th of that per rank; four times the even share absorbs what a picture does to the
        // distribution.  (An overflow is detected, reported by wfx_shard_result and cured by a larger capacity.)
        uint64_t want = p->n / (512ull * (uint64_t)(pl.world > 0 ? pl.world : 1));
        // (plan 3 keeps everything else on the wire in kilobytes: its floor is a quarter of the transposing plans' -- an overflow is
        // reported and cured like theirs)
        uint64_t cap = pl.fmm ? SH_CAND_CAP / 4 : SH_CAND_CAP;
        while (cap < want && cap < (1ull << 20)) cap *= 2;
        sh->cap = cap;
    }
    if (rc == 0) rc = wfx_reserve(ctx, ctx->b_scal, sizeof(wfx_dev_scalars));
    // The plan fixes the sequence and sizes of the collectives, and part of it comes from each process's OWN environment (WFX_LINK_GBS,
    // WFX_LINK_LAT_US, WFX_SHARD_CHUNKS, WFX_SHARD_ROWS): a rank started with another environment would issue other exchanges, and
    // RCCL then hangs instead of reporting anything.  The ranks compare a digest of what they decided before any of it is used
    // (real multi-process transports only: the ranks of an in-process world share one environment and are created one by one).
    if (rc == 0 && pl.world > 1 && !wfx_comm_is_local(comm)) {
        long long dig[8] = {pl.single ? 1 : (pl.fmm ? 3 : 0), pl.cols ? 1 : 0, (long long)pl.nchunk, (long long)pl.g.R1, (long long)pl.Kp, (long long)pl.M1,
                            (long long)p->n0, (long long)p->n};
        wfx_devbuf &b = sh->b_flags;
        rc = wfx_reserve(ctx, b, 64 + (size_t)pl.world * 64);
        std::vector<long long> all((size_t)pl.world * 8, -1);
        if (rc == 0 && hipMemcpyAsync(b.p, dig, 64, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = wfx_fail(ctx, WFX_ERR_HIP, "plan digest upload");
        if (rc == 0) {
            wfx_comm_label(comm, "plan digest");
            rc = wfx_comm_allgather(comm, ctx, b.p, (char *)b.p + 64, 64);
        }
        if (rc == 0 && (hipMemcpyAsync(all.data(), (char *)b.p + 64, (size_t)pl.world * 64, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                        hipStreamSynchronize(ctx->stream) != hipSuccess))
            rc = wfx_fail(ctx, WFX_ERR_HIP, "plan digest download");
        for (int r = 0; rc == 0 && r < pl.world; ++r)
            for (int k = 0; k < 8; ++k)
                if (all[(size_t)r * 8 + k] != dig[k]) {
                    static const char *what[8] = {"single plan", "columns layout", "k1 subsets", "first radix", "Hilbert length", "resampler length", "input frames", "samples"};
                    rc = wfx_fail(ctx, WFX_ERR_COMM, "sharded decode: rank %d decided %s = %lld, rank %d %lld -- the ranks' environments (WFX_LINK_GBS, "
                                  "WFX_LINK_LAT_US, WFX_SHARD_CHUNKS, WFX_SHARD_ROWS) or parameters differ", r, what[k], all[(size_t)r * 8 + k], pl.rank, dig[k]);
                    break;
                }
        if (rc != 0) {
            wfx_shard_destroy(sh);
            return rc;
        }
    }
    if (rc == 0 && pl.single) {
        rc = wfx_reserve(ctx, sh->b_flags, 64);
        if (rc == 0 && hipMemsetAsync(sh->b_flags.p, 0, 64, ctx->stream) != hipSuccess) rc = wfx_fail(ctx, WFX_ERR_HIP, "memset");
        if (rc != 0) {
            wfx_shard_destroy(sh);
            return rc;
        }
        *out = sh;
        return 0;
    }
    if (rc == 0 && pl.fmm) {      // plan 3: no distributed transform objects
        *out = sh;
        return 0;
    }
    sh->dF.set_tag("resample fwd");
    sh->dI.set_tag("resample inv");
    sh->dH.set_tag("hilbert");
    sh->dHk.set_tag("hilbert kernel");
    if (rc == 0 && pl.resample) {
        const int nc = pl.cols ? pl.nchunk : 1;
        rc = sh->dF.init(ctx, pl.g, pl.M1, pl.in_kind == WFX_IN_I16_MONO ? 4 : 16, 0, 0, false, 0, nc);
        if (rc == 0) rc = sh->dI.init(ctx, pl.g, pl.K, 16, pl.cols ? pl.hs / 2 : SH_HALO / 2, pl.cols ? pl.hs / 2 : SH_HALO / 2, false, 0, nc);
    }
    if (rc == 0) rc = sh->dH.init(ctx, pl.g, pl.Kp, 16, SH_VHALO, SH_VHALO, false, 0, pl.cols ? pl.nchunk : 1);
    if (rc == 0 && pl.split_kernel) rc = sh->dHk.init(ctx, pl.gk, pl.Kp, 16, 0, 0);
    if (rc != 0) {
        wfx_shard_destroy(sh);
        return rc;
    }
    *out = sh;
    return 0;
}

int wfx_shard_upload(wfx_shard *sh, const void *host_frames)
{
    CHECK_SH(sh);
    wfx_ctx *ctx = sh->ctx;
    const size_t nb = (size_t)(sh->pl.cols ? cols_in_frames(sh->pl) : sh->pl.in_hi - sh->pl.in_lo + (sh->pl.fmm ? 2 * sh->pl.in_halo : 0)) * frame_bytes(sh->pl.in_kind);
    if (!host_frames && nb) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
    const bool moved = !sh->b_in.p || sh->b_in.cap < nb + 64 || sh->ext_in;
    WFX_TRY(wfx_reserve(ctx, sh->b_in, nb + 64));
    sh->ext_in = nullptr;
    if (nb) WFX_HIP(ctx, hipMemcpyAsync(sh->b_in.p, host_frames, nb, hipMemcpyHostToDevice, ctx->stream));
    WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    sh->have_input = true;
    sh->ran = false;
    if (moved || !sh->bound) WFX_TRY(shard_bind(sh));
    return 0;
}

int wfx_shard_attach(wfx_shard *sh, const void *dev_frames)
{
    CHECK_SH(sh);
    if (!dev_frames) return wfx_fail(sh->ctx, WFX_ERR_BAD_ARG, "null buffer");
    const bool moved = sh->ext_in != dev_frames;
    sh->ext_in = dev_frames;
    sh->have_input = true;
    sh->ran = false;
    if (moved || !sh->bound) WFX_TRY(shard_bind(sh));
    return 0;
}

int wfx_shard_phase_count(wfx_shard *sh)
{
    if (!sh) return 0;
    settle_kernel(sh);
    return phase_count(sh);
}

int wfx_shard_phase(wfx_shard *sh, int phase)
{
    CHECK_SH(sh);
    if (!sh->have_input) return wfx_fail(sh->ctx, WFX_ERR_STATE, "sharded decode before the input was given");
    if (phase == 0) {                                    // a new decode (whatever became of the previous one)
        sh->in_decode = false;
        settle_kernel(sh);
    }
    const int np = phase_count(sh);
    if (phase < 0 || phase >= np) return wfx_fail(sh->ctx, WFX_ERR_BAD_ARG, "phase %d out of range", phase);
    sh->in_decode = true;
    const int rc = run_phase(sh, phase);
    if (rc != 0) {
        sh->in_decode = false;
        return rc;
    }
    if (phase == np - 1) {
        sh->in_decode = false;
        sh->ran = true;
        if (sh->pl.padded) {                             // (the next decode has three phases fewer)
            sh->ghat_ready = true;
            sh->ghat_pending = false;
        }
    }
    return 0;
}

int wfx_decode_sharded(wfx_shard *sh)
{
    CHECK_SH(sh);
    if (wfx_comm_is_local(sh->comm) && wfx_comm_world(sh->comm) > 1)
        return wfx_fail(sh->ctx, WFX_ERR_STATE, "local communicator: drive the ranks phase by phase (wfx_shard_phase)");
    const int np = wfx_shard_phase_count(sh);
    for (int ph = 0; ph < np; ++ph) WFX_TRY(wfx_shard_phase(sh, ph));
    return 0;
}

int wfx_shard_result(wfx_shard *sh, wfx_decode_info *info)
{
    CHECK_SH(sh);
    wfx_ctx *ctx = sh->ctx;
    if (!info) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null info");
    if (!sh->ran) return wfx_fail(ctx, WFX_ERR_STATE, "shard_result before the decode");
    if (sh->pl.single && sh->pl.rank == 0) return wfx_decode_result(ctx, info);
    wfx_dev_scalars *ds = (wfx_dev_scalars *)ctx->b_scal.p;
    for (int attempt = 0;; ++attempt) {
        unsigned flags[4] = {0, 0, 0, 0};
        WFX_HIP(ctx, hipMemcpyAsync(flags, sh->b_flags.p, sizeof flags, hipMemcpyDeviceToHost, ctx->stream));
        if (sh->pl.rank != 0) WFX_HIP(ctx, hipMemcpyAsync(ctx->h_scal, ds, sizeof(wfx_dev_scalars), hipMemcpyDeviceToHost, ctx->stream));
        WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (!flags[0]) break;
        // A rank produced more candidate keys than travel in the all-gather (long runs of equal envelope values: digital silence,
        // a clipped carrier).  The flag comes out of the merged blocks, which are identical on every rank: every rank takes this
        // branch in the same decode.  The capacity grows 16x and the capture is decoded again -- here, when this process can run
        // the phases by itself (RCCL / shared-memory transport, or one rank); the ranks of an in-process world are driven phase by
        // phase by their caller, who gets the same message on every rank and repeats the decode (sharded.decode_emulated does).
        WFX_HIP(ctx, hipMemsetAsync(sh->b_flags.p, 0, 64, ctx->stream));
        sh->cap *= 16;
        sh->bound = false;
        sh->ran = false;
        WFX_TRY(shard_bind(sh));
        const bool self_driven = !(wfx_comm_is_local(sh->comm) && wfx_comm_world(sh->comm) > 1);
        if (!self_driven || attempt >= 6 || sh->cap > (1ull << 34))
            return wfx_fail(ctx, WFX_ERR_STATE, "percentile select: candidate lists overflowed; capacity raised to %llu keys, decode again",
                            (unsigned long long)sh->cap);
        WFX_TRY(wfx_decode_sharded(sh));
    }
    const wfx_dev_scalars &s = *ctx->h_scal;
    memset(info, 0, sizeof *info);
    info->n = sh->pl.n;
    info->width = sh->dp.width;
    info->low = s.low;
    info->high = s.high;
    if (sh->pl.rank == 0) {
        info->nan_count = s.nan_count;
        info->npeaks = s.npeaks;
        info->hit_limit = s.hit_limit;
        info->no_group = s.no_group;
        info->n_phasing = s.n_phasing;
        info->start_frame = s.start_frame;
        info->height = s.height;
        for (int i = 0; i <= WFX_MAX_PEAKS; ++i) {
            info->peak_pos[i] = s.peak_pos[i];
            info->first_pos[i] = s.first_pos[i];
            info->phasing[i] = s.phasing[i];
        }
    }
    return 0;
}

int wfx_shard_fetch(wfx_shard *sh, int buffer_id, void *host_out, size_t bytes)
{
    CHECK_SH(sh);
    wfx_ctx *ctx = sh->ctx;
    if (!sh->ran) return wfx_fail(ctx, WFX_ERR_STATE, "no sharded decode has run");
    const shard_plan &pl = sh->pl;
    const uint64_t n_own = pl.own_hi - pl.own_lo;
    if (pl.single) {
        if (pl.rank == 0) return wfx_decode_fetch(ctx, buffer_id, host_out, bytes);
        if (buffer_id == WFX_BUF_IMAGE) return wfx_fail(ctx, WFX_ERR_STATE, "the image lives on rank 0");
        if (bytes != 0) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "fetch: expected 0 bytes, got %zu (rank 0 owns the whole capture)", bytes);
        return 0;
    }
    if (pl.cols) {
        const uint64_t nc = cols_own(pl);
        const void *src = nullptr;
        size_t nb = 0;
        switch (buffer_id) {
        case WFX_BUF_AUDIO:       // the own samples of the R1 rows, back to back
            if (bytes != nc * 8) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "fetch: expected %zu bytes, got %zu", (size_t)nc * 8, bytes);
            if (!host_out) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
            WFX_HIP(ctx, hipMemcpy2DAsync(host_out, (size_t)pl.wH * 16, (const double *)sh->b_audio.p + SH_PAD + pl.hs, (size_t)pl.xrs * 8, (size_t)pl.wH * 16,
                                          (size_t)pl.g.R1, hipMemcpyDeviceToHost, ctx->stream));
            WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
            return 0;
        case WFX_BUF_ENVELOPE: src = sh->b_env.p; nb = nc * 8; break;
        case WFX_BUF_DIGITAL:
            if (pl.rank == 0 && bytes == pl.n && (pl.world > 1 || nc != pl.n)) {
                src = sh->b_dig.p;
                nb = pl.n;
            } else {
                src = (pl.rank == 0 && pl.world > 1) ? sh->b_gath.p : sh->b_dig.p;
                nb = nc;
            }
            break;
        case WFX_BUF_IMAGE:
            if (pl.rank != 0) return wfx_fail(ctx, WFX_ERR_STATE, "the image lives on rank 0");
            WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
            src = ctx->b_img.p;
            nb = (size_t)sh->dp.width * 4 * (size_t)ctx->h_scal->height;
            break;
        default: return wfx_fail(ctx, WFX_ERR_BAD_ARG, "unknown buffer id %d", buffer_id);
        }
        if (bytes != nb) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "fetch: expected %zu bytes, got %zu", nb, bytes);
        if (nb && !host_out) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
        if (nb) WFX_HIP(ctx, hipMemcpyAsync(host_out, src, nb, hipMemcpyDeviceToHost, ctx->stream));
        WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return 0;
    }
    const void *src = nullptr;
    size_t nb = 0;
    switch (buffer_id) {
    case WFX_BUF_AUDIO: src = (const double *)sh->b_audio.p + (pl.fmm ? (uint64_t)SH_FMM_HF : pl.own_lo - pl.seg_lo); nb = n_own * 8; break;
    case WFX_BUF_ENVELOPE: src = sh->b_env.p; nb = n_own * 8; break;
    case WFX_BUF_DIGITAL:
        if (pl.rank == 0 && bytes == pl.n) {
            src = sh->b_dig.p;
            nb = pl.n;
        } else {
            src = (const uint8_t *)sh->b_dig.p + (pl.rank == 0 ? pl.own_lo : 0);
            nb = n_own;
        }
        break;
    case WFX_BUF_IMAGE:
        if (pl.rank != 0) return wfx_fail(ctx, WFX_ERR_STATE, "the image lives on rank 0");
        WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        src = ctx->b_img.p;
        nb = (size_t)sh->dp.width * 4 * (size_t)ctx->h_scal->height;
        break;
    default: return wfx_fail(ctx, WFX_ERR_BAD_ARG, "unknown buffer id %d", buffer_id);
    }
    if (bytes != nb) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "fetch: expected %zu bytes, got %zu", nb, bytes);
    if (nb && !host_out) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
    if (nb) WFX_HIP(ctx, hipMemcpyAsync(host_out, src, nb, hipMemcpyDeviceToHost, ctx->stream));
    WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int wfx_shard_destroy(wfx_shard *sh)
{
    if (!sh) return 0;
    if (!wfx_ctx_alive(sh->ctx)) {          // its context went first: the device memory went with the process's teardown or leaks
        delete sh;
        return 0;
    }
    (void)hipSetDevice(sh->ctx->device);
    (void)hipStreamSynchronize(sh->ctx->stream);
    sh->dF.release();
    sh->dI.release();
    sh->dH.release();
    sh->dHk.release();
    wfx_devbuf *bufs[] = {&sh->b_in, &sh->b_merged, &sh->b_res, &sh->b_audio, &sh->b_v, &sh->b_env, &sh->b_dig, &sh->b_blk, &sh->b_blks, &sh->b_nan, &sh->b_flags,
                          &sh->b_grow, &sh->b_ghat, &sh->b_gath, &sh->b_pieces};
    for (wfx_devbuf *b : bufs) free_buf(*b);
    delete sh;
    return 0;
}

}  // extern "C"
