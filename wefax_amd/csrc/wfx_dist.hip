// Distributed exact transforms of the sharded decode: see wfx_dist.h for the layouts.
//
// Why this exists: the two operators that make the reference's path global -- scipy.signal.hilbert (wefax.py:174) and
// scipy.signal.resample (wefax.py:384) -- are DFTs over the whole capture.  Truncating them to halo-local FIRs costs
// +-1 LSB (SURVEY.md appendix B.2, round-1 measurements), which moves sync peaks and with them the whole image.  A
// Cooley-Tukey split L = R1 * M keeps them exact on N GPUs: the first pass is R1-point transforms along n1 (needs all n1 of a
// column -> "columns" layout), the rest is R1 independent M-point transforms (needs all n2 of a first-pass output k1 ->
// "slab" layout).  Two transposes per transform are the real exchange step of the path; everything else stays local.
#include <algorithm>

#include "wfx_dist.h"

// ---- geometry (host) ---------------------------------------------------------------------------------------
bool wfx_dist_make_geom(wfx_dist_geom &g, int world, int rank, int ra1, int rb1, int rows_used)
{
    if (world < 1 || rank < 0 || rank >= world || !wfx_mr_is_pair(ra1, rb1)) return false;
    g.world = world;
    g.rank = rank;
    g.ra1 = ra1;
    g.rb1 = rb1;
    g.R1 = ra1 * rb1;
    const int R1 = g.R1, units = R1 / 2 + 1;
    if (units < 2 * world) return false;              // every rank needs >= 2 first-pass outputs and >= 1 row
    g.rows_used = (rows_used > 0 && rows_used < R1) ? rows_used : R1;
    if (g.rows_used < world) return false;
    g.rows.resize(world + 1);
    for (int s = 0; s <= world; ++s) g.rows[s] = (int)((long long)s * g.rows_used / world);
    g.km.resize(world);
    for (int e = 0; e < world; ++e) {
        // units u in [a, b) stand for {u, R1 - u}: the set is closed under k1 -> -k1, so the bins k and L - k of a real
        // signal's spectrum sit on the same rank (the resampler's untangling needs both)
        const int a = (int)((long long)e * units / world), b = (int)((long long)(e + 1) * units / world);
        int lo1 = R1 - b + 1, hi1 = R1 - a + 1;
        if (a == 0) hi1 = R1;                                      // 0 is its own mirror
        if (R1 % 2 == 0 && b - 1 == R1 / 2) lo1 = R1 / 2 + 1;      // so is R1 / 2
        if (hi1 < lo1) hi1 = lo1;
        g.km[e].kb0 = a;
        g.km[e].kc0 = b - a;
        g.km[e].kb1 = lo1;
        g.km[e].B = (b - a) + (hi1 - lo1);
        if (g.km[e].B < 2) return false;
    }
    return true;
}

bool wfx_dist_choose_r1(const long long *lengths, int nlen, int world, int *ra1, int *rb1, bool pairs_required)
{
    // largest first radix that divides every length, leaves pair-decomposable cofactors and gives every rank work:
    // a large R1 balances rows / slabs over the ranks and keeps the first pass's tiles wide
    std::vector<std::pair<int, int>> cand;
    wfx_mr_all_pairs(cand);
    std::stable_sort(cand.begin(), cand.end(), [](const std::pair<int, int> &x, const std::pair<int, int> &y) { return x.first * x.second > y.first * y.second; });
    // first choice: every remaining pass a register-resident radix pair; otherwise any 13-smooth cofactor (per-prime passes)
    for (int pairs_only = 1; pairs_only >= (pairs_required ? 1 : 0); --pairs_only)
        for (const auto &c : cand) {
            const int R1 = c.first * c.second;
            wfx_dist_geom g;
            if (!wfx_dist_make_geom(g, world, 0, c.first, c.second)) continue;
            bool ok = true;
            for (int i = 0; i < nlen && ok; ++i) {
                std::vector<wfx_mr_radix> sub;
                std::vector<std::pair<int, int>> prs;
                ok = lengths[i] % R1 == 0 && lengths[i] / R1 >= 4ll * world &&
                     (pairs_only ? wfx_mr_pair_plan(lengths[i] / R1, prs) : wfx_mr_general_plan(lengths[i] / R1, sub));
            }
            if (ok) {
                *ra1 = c.first;
                *rb1 = c.second;
                return true;
            }
        }
    return false;
}

// ---- copies around an exchange --------------------------------------------------------------------------
__device__ __forceinline__ int dist_k1(const wfx_dist_piece &p, int kk) { return kk < p.kc0 ? p.kb0 + kk : p.kb1 + (kk - p.kc0); }

template <typename T>
__global__ void __launch_bounds__(256) dist_copy2d_kernel(const wfx_dist_piece *__restrict__ pieces)
{
    const wfx_dist_piece p = pieces[blockIdx.y];
    const T *__restrict__ src = (const T *)p.src;
    T *__restrict__ dst = (T *)p.dst;
    const unsigned total = (unsigned)p.rows * (unsigned)p.cols, cols = (unsigned)p.cols;
    for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < total; e += gridDim.x * 256u) {
        const unsigned r = e / cols, c = e - r * cols;
        dst[(long long)r * p.dst_rs + c] = src[(long long)r * p.src_rs + c];
    }
}

// dst[k1(kk) dst_rs + j] = src[j B + kk]: a slab's rows back into the [R1][w] layout of the last inverse pass
// (a transposition: 64 x 16 tiles through LDS, 256-byte reads, 1-KB writes)
__global__ void __launch_bounds__(256) dist_scatter_k_kernel(const wfx_dist_piece *__restrict__ pieces)
{
    __shared__ cplx tile[64][17];
    const wfx_dist_piece p = pieces[blockIdx.y];
    const cplx *__restrict__ src = (const cplx *)p.src;
    cplx *__restrict__ dst = (cplx *)p.dst;
    const int t = threadIdx.x, B = p.B;
    for (int j0 = blockIdx.x * 64; j0 < p.rows; j0 += gridDim.x * 64) {
        for (int kk0 = 0; kk0 < B; kk0 += 16) {
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int k = t & 15, jj = (t >> 4) + 16 * it;
                if (j0 + jj < p.rows && kk0 + k < B) tile[jj][k] = src[(long long)(j0 + jj) * B + kk0 + k];
            }
            __syncthreads();
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int jj = t & 63, k = (t >> 6) + 4 * it;
                if (j0 + jj < p.rows && kk0 + k < B) dst[(long long)dist_k1(p, kk0 + k) * p.dst_rs + j0 + jj] = tile[jj][k];
            }
        }
    }
}

static unsigned copy_grid(long long max_elems)
{
    long long b = (max_elems + 2047) / 2048;       // ~8 elements per thread
    if (b < 1) b = 1;
    if (b > 1024) b = 1024;
    return (unsigned)b;
}

int wfx_dist_copy2d(wfx_ctx *ctx, const wfx_dist_piece *dev_pieces, int npieces, long long max_elems, int elem_bytes)
{
    if (npieces <= 0 || max_elems <= 0) return 0;
    if (elem_bytes == 16)
        WFX_LAUNCH(ctx, K_DIST_COPY, dist_copy2d_kernel<double2>, dim3(copy_grid(max_elems), npieces), dim3(256), dev_pieces);
    else if (elem_bytes == 8)
        WFX_LAUNCH(ctx, K_DIST_COPY, dist_copy2d_kernel<double>, dim3(copy_grid(max_elems), npieces), dim3(256), dev_pieces);
    else if (elem_bytes == 4)
        WFX_LAUNCH(ctx, K_DIST_COPY, dist_copy2d_kernel<unsigned>, dim3(copy_grid(max_elems), npieces), dim3(256), dev_pieces);
    else if (elem_bytes == 2)
        WFX_LAUNCH(ctx, K_DIST_COPY, dist_copy2d_kernel<unsigned short>, dim3(copy_grid(max_elems), npieces), dim3(256), dev_pieces);
    else
        return wfx_fail(ctx, WFX_ERR_BAD_ARG, "copy2d: element size %d", elem_bytes);
    return 0;
}

int wfx_dist_scatter_k(wfx_ctx *ctx, const wfx_dist_piece *dev_pieces, int npieces, int max_rows)
{
    if (npieces <= 0 || max_rows <= 0) return 0;
    unsigned gx = (unsigned)((max_rows + 63) / 64);
    if (gx > 2048) gx = 2048;
    WFX_LAUNCH(ctx, K_DIST_COPY, dist_scatter_k_kernel, dim3(gx, npieces), dim3(256), dev_pieces);
    return 0;
}

// ---- one distributed transform length -------------------------------------------------------------------
void wfx_dist::buffers(std::vector<std::pair<unsigned long long, unsigned long long>> &out) const
{
    const wfx_devbuf *bufs[] = {&b_pack, &b_recv, &b_y, &b_a, &b_a2, &b_halo};
    for (const wfx_devbuf *b : bufs)
        if (b->p) out.emplace_back((unsigned long long)b->p, (unsigned long long)b->cap);
}

void wfx_dist::label(wfx_comm *c, const char *e) const
{
    char b[24];
    snprintf(b, sizeof b, "%.15s %s", tag_, e);
    wfx_comm_label(c, b);
}

void wfx_dist::release()
{
    wfx_devbuf *bufs[] = {&tables, &b_pack, &b_recv, &b_y, &b_a, &b_a2, &b_desc, &b_halo};
    for (wfx_devbuf *b : bufs) {
        if (b->p && !dry_) (void)hipFree(b->p);
        b->p = nullptr;
        b->cap = 0;
    }
}

int wfx_dist::init(wfx_ctx *ctx_, const wfx_dist_geom &g_, long long L_, int elem_bytes_in, int halo_before, int halo_after, bool dry,
                   unsigned long long dry_base, int nchunk)
{
    ctx = ctx_;
    dry_ = dry;
    g = g_;
    L = L_;
    es_in = elem_bytes_in;
    hb = halo_before;
    ha = halo_after;
    const int R1 = g.R1, W = g.world, me = g.rank;
    if (L % R1) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "distributed transform: %lld is not a multiple of the first radix %d", L, R1);
    M = L / R1;
    if (L >= (1ll << 31)) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "distributed transform: %lld points exceed 32-bit indices", L);
    if (!wfx_mr_general_plan(M, sub)) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "distributed transform: %lld is not 13-smooth", M);
    cols.resize(W + 1);
    for (int d = 0; d <= W; ++d) cols[d] = d == W ? M : (long long)d * M / W / 4 * 4;
    w = (int)(cols[me + 1] - cols[me]);
    B = g.km[me].B;
    nr = g.nrows(me);
    for (int d = 0; d < W; ++d)
        if (cols[d + 1] - cols[d] < 2 || cols[d + 1] - cols[d] < std::max(hb, ha))
            return wfx_fail(ctx, WFX_ERR_BAD_ARG, "distributed transform: %lld points are too few for %d ranks", L, W);
    if (nr < 1) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "distributed transform: rank %d owns no rows", me);

    // ---- subsets of the k1 set: the k1 sets of a geometry with W * C ranks (rank r's subsets are the virtual ranks r C .. r C + C - 1:
    // the unit ranges nest, so together they are exactly rank r's own set) ----
    C = nchunk < 1 ? 1 : nchunk;
    wfx_dist_geom gv;
    while (C > 1 && !wfx_dist_make_geom(gv, W * C, me * C, g.ra1, g.rb1)) --C;
    kmv.clear();
    if (C > 1) {
        kmv = gv.km;
        int tot = 0;
        for (int c = 0; c < C; ++c) tot += kmv[(size_t)me * C + c].B;
        if (tot != B || kmv[(size_t)me * C].kb0 != g.km[me].kb0) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "distributed transform: the k1 subsets do not tile the rank's set");
    } else
        kmv = g.km;
    kmc.assign(kmv.begin() + (size_t)me * C, kmv.begin() + (size_t)me * C + C);
    soff.assign(C + 1, 0);
    for (int c = 0; c < C; ++c) soff[c + 1] = soff[c] + M * (long long)kmc[c].B;
    // ---- pass descriptors and twiddle tables ----
    const int ns = (int)sub.size();
    size_t off = 0;
    wfx_mr_pair_desc(d_first, g.ra1, g.rb1, 1, w, (long long)R1 * w);
    d_fwd.assign(C, std::vector<mr_pass_desc>(ns));
    d_inv.assign(C, std::vector<mr_pass_desc>(ns));
    tw_fwd.resize(ns);
    tw_inv.resize(ns);
    {
        long long Ptw = R1;
        for (int i = 0; i < ns; ++i) {              // forward: the global plan's passes 2.., twiddles by global k = k1 + R1 (.)
            tw_fwd[i] = off;
            off += wfx_mr_table_elems(Ptw * sub[i].R);
            Ptw *= sub[i].R;
        }
        long long Psub = 1;
        for (int i = 0; i < ns; ++i) {              // inverse: M-point transforms, radices in the reverse order
            tw_inv[i] = off;
            off += wfx_mr_table_elems(Psub * sub[ns - 1 - i].R);
            Psub *= sub[ns - 1 - i].R;
        }
    }
    for (int c = 0; c < C; ++c) {
        const int Bc = kmc[c].B;
        long long Play = Bc, Ptw = R1;
        for (int i = 0; i < ns; ++i) {
            const int R = sub[i].R;
            mr_pass_desc &d = d_fwd[c][i];
            wfx_mr_general_desc(d, sub[i], Play, (long long)Bc * M / R, (long long)Bc * M);
            d.dist = 1;
            d.B = Bc;
            d.kb0 = kmc[c].kb0;
            d.kc0 = kmc[c].kc0;
            d.kb1 = kmc[c].kb1;
            d.kscale = 1;
            d.kstep = R1;
            d.Ptw = Ptw;
            d.Ltw = L;
            Play *= R;
            Ptw *= R;
        }
        Play = Bc;
        long long Psub = 1;
        for (int i = 0; i < ns; ++i) {
            const wfx_mr_radix pr = sub[ns - 1 - i];
            const int R = pr.R;
            mr_pass_desc &d = d_inv[c][i];
            wfx_mr_general_desc(d, pr, Play, (long long)Bc * M / R, (long long)Bc * M);
            d.dist = 1;
            d.B = Bc;
            d.kb0 = d.kb1 = 0;
            d.kc0 = Bc;
            d.kscale = 0;
            d.kstep = 1;
            d.Ptw = Psub;
            d.Ltw = L;
            Play *= R;
            Psub *= R;
        }
    }
    wfx_mr_pair_desc(d_last, g.ra1, g.rb1, w, w, (long long)R1 * w);
    d_last.dist = 1;
    d_last.B = 1;
    d_last.kb0 = (int)cols[me];
    d_last.kc0 = 1;
    d_last.kb1 = 0;
    d_last.kscale = 1;
    d_last.kstep = 1;
    d_last.Ptw = M;
    d_last.Ltw = L;
    tw_last = off;
    off += wfx_mr_table_elems(L);
    const size_t colsz = (size_t)R1 * w, rowsz = (size_t)nr * M, slab = (size_t)M * B;
    const size_t halosz = (size_t)R1 * (size_t)(hb + ha) * 2 * sizeof(cplx) + 64;      // columns mode: halo columns out and in
    if (dry) {
        b_halo.p = (void *)(dry_base + (8ull << 36));
        b_halo.cap = halosz;
        wfx_devbuf *bufs[] = {&b_pack, &b_recv, &b_y, &b_a, &b_a2};
        const size_t caps[] = {std::max(colsz, rowsz) * sizeof(cplx) + 64, std::max(colsz, rowsz) * sizeof(cplx) + 64, colsz * sizeof(cplx) + 64,
                               slab * sizeof(cplx) + 64, slab * sizeof(cplx) + 64};
        for (int i = 0; i < 5; ++i) {
            bufs[i]->p = (void *)(dry_base + ((unsigned long long)(i + 1) << 36));
            bufs[i]->cap = caps[i];
        }
        last_rows_in = nullptr;
        last_rows_out = nullptr;
        return 0;
    }
    WFX_TRY(wfx_reserve(ctx, tables, off * sizeof(cplx)));
    cplx *tb = (cplx *)tables.p;
    {
        long long Ptw = R1;
        for (int i = 0; i < ns; ++i) {
            const int R = sub[i].R;
            WFX_TRY(wfx_mr_fill_table(ctx, tb + tw_fwd[i], Ptw * R));
            Ptw *= R;
        }
        long long Psub = 1;
        for (int i = 0; i < ns; ++i) {
            const int R = sub[ns - 1 - i].R;
            WFX_TRY(wfx_mr_fill_table(ctx, tb + tw_inv[i], Psub * R));
            Psub *= R;
        }
        WFX_TRY(wfx_mr_fill_table(ctx, tb + tw_last, L));
    }
    // ---- buffers ----
    WFX_TRY(wfx_reserve(ctx, b_pack, std::max(colsz, rowsz) * sizeof(cplx) + 64));
    WFX_TRY(wfx_reserve(ctx, b_recv, std::max(colsz, rowsz) * sizeof(cplx) + 64));
    WFX_TRY(wfx_reserve(ctx, b_y, colsz * sizeof(cplx) + 64));
    WFX_TRY(wfx_reserve(ctx, b_a, slab * sizeof(cplx) + 64));
    WFX_TRY(wfx_reserve(ctx, b_a2, slab * sizeof(cplx) + 64));
    WFX_TRY(wfx_reserve(ctx, b_halo, halosz));
    last_rows_in = nullptr;
    last_rows_out = nullptr;
    return 0;
}

// Exchange lists and piece descriptors.  They hold absolute addresses, so they are rebuilt when the caller's row buffers move.
void wfx_dist::build_lists(const void *rows_in, cplx *rows_out)
{
    const int W = g.world, me = g.rank, R1 = g.R1;
    const size_t ES = (size_t)es_in;
    char *pack = (char *)b_pack.p, *recv = (char *)b_recv.p;
    cplx *Y = (cplx *)b_y.p, *A = (cplx *)b_a.p;
    x1.clear(); x2.clear(); x3.clear(); x4.clear(); xh.clear();
    p1.clear(); p3.clear(); p4.clear(); ph.clear();
    nph_pack = 0;
    qmap.assign(R1, mr_qmap{0, 0});
    // one rank: the rows ARE the columns, so the first pass reads the caller's rows and the last inverse pass writes the caller's
    // output directly; the exchanges E1 / E4 then carry nothing but the (self) halo copies
    pass1_src = (W == 1 || cols_) ? rows_in : (const void *)recv;
    pass_last_dst = ((W == 1 || cols_) && rows_out) ? rows_out + hb : (cplx *)pack;
    // E1: rows -> columns
    if (rows_in && W > 1 && !cols_) {
        size_t off = 0;
        for (int d = 0; d < W; ++d) {
            const long long wd = cols[d + 1] - cols[d];
            char *self_dst = recv + (size_t)g.rows[me] * w * ES;
            char *dst = d == me ? self_dst : pack + off;
            wfx_dist_piece p{};
            p.src = (unsigned long long)((const char *)rows_in + (size_t)cols[d] * ES);
            p.dst = (unsigned long long)dst;
            p.rows = nr;
            p.cols = (int)wd;
            p.src_rs = M;
            p.dst_rs = wd;
            p1.push_back(p);
            wfx_xfer x{};
            x.peer = d;
            x.send = dst;
            x.send_bytes = (size_t)nr * wd * ES;
            x.recv = recv + (size_t)g.rows[d] * w * ES;
            x.recv_bytes = (size_t)g.nrows(d) * w * ES;
            x1.push_back(x);
            if (d != me) off += (size_t)nr * wd * ES;
        }
    }
    // E2: first-pass output (column j, output k1) -> slabs [M][B_(e,c)], one per rank e and subset c.  No packing copy: the first
    // pass stores every output where the exchange sends it from (mr_pass_desc::qmap), the own part straight into the slab.
    x2c.assign(C, {});
    x3c.assign(C, {});
    {
        size_t off = 0;
        for (int c = 0; c < C; ++c)
            for (int e = 0; e < W; ++e) {
                const wfx_dist_kmap &km = kmv[(size_t)e * C + c];
                const int Bme = kmc[c].B;
                cplx *self_dst = A + soff[c] + (size_t)cols[me] * Bme;
                cplx *dst = e == me ? self_dst : (cplx *)pack + off;
                for (int kk = 0; kk < km.B; ++kk) {
                    const int k1 = kk < km.kc0 ? km.kb0 + kk : km.kb1 + (kk - km.kc0);
                    qmap[k1].base = (unsigned long long)(dst + kk);
                    qmap[k1].stride = km.B;
                }
                wfx_xfer x{};
                x.peer = e;
                x.send = dst;
                x.send_bytes = (size_t)w * km.B * sizeof(cplx);
                x.recv = A + soff[c] + (size_t)cols[e] * Bme;
                x.recv_bytes = (size_t)(cols[e + 1] - cols[e]) * Bme * sizeof(cplx);
                x2c[c].push_back(x);
                x2.push_back(x);
                if (e != me) off += (size_t)w * km.B;
            }
    }
    // E3: slab rows -> [R1][w] (S = inv_result, known after init: see inv_slab_exchange)
    {
        cplx *S = inv_result;
        size_t off = 0;
        for (int c = 0; c < C; ++c)
            for (int e = 0; e < W; ++e) {
                const wfx_dist_kmap &km = kmv[(size_t)e * C + c];
                const int Bme = kmc[c].B;
                cplx *Sc = S + soff[c];
                cplx *src = e == me ? Sc + (size_t)cols[me] * Bme : (cplx *)recv + off;
                wfx_dist_piece p{};
                p.src = (unsigned long long)src;
                p.dst = (unsigned long long)Y;
                p.rows = w;
                p.cols = km.B;
                p.src_rs = km.B;
                p.dst_rs = w;
                p.kb0 = km.kb0;
                p.kc0 = km.kc0;
                p.kb1 = km.kb1;
                p.B = km.B;
                p3.push_back(p);
                wfx_xfer x{};
                x.peer = e;
                x.send = Sc + (size_t)cols[e] * Bme;                    // to rank e: this subset's slab rows of its columns (contiguous)
                x.send_bytes = (size_t)(cols[e + 1] - cols[e]) * Bme * sizeof(cplx);
                x.recv = src;
                x.recv_bytes = (size_t)w * km.B * sizeof(cplx);
                if (e == me) x.send = x.recv;
                x3c[c].push_back(x);
                x3.push_back(x);
                if (e != me) off += (size_t)w * km.B;
            }
    }
    // columns mode: no E4 -- the rows stay where the last pass wrote them (rows_out + q out_rs + hb) and only the halo columns
    // travel: my last hb own columns of every row to the rank on my right (its "before" halo), my first ha own columns to the
    // rank on my left (its "after" halo); across the wrap (last rank -> rank 0 and back) the rows shift by one, circularly
    if (rows_out && cols_ && (hb > 0 || ha > 0)) {
        cplx *X = rows_out;                                           // row q at X + q out_rs: [hb | w | ha]
        cplx *H = (cplx *)b_halo.p;
        cplx *sendR = H, *sendL = H + (size_t)R1 * hb, *recvL = H + (size_t)R1 * (hb + ha), *recvR = recvL + (size_t)R1 * hb;
        const int right = (me + 1) % W, left = (me + W - 1) % W;
        auto piece = [&](const cplx *src, cplx *dst, int rows, int cols_n, long long src_rs, long long dst_rs) {
            wfx_dist_piece p{};
            p.src = (unsigned long long)src;
            p.dst = (unsigned long long)dst;
            p.rows = rows;
            p.cols = cols_n;
            p.src_rs = src_rs;
            p.dst_rs = dst_rs;
            if (rows > 0 && cols_n > 0) ph.push_back(p);
        };
        if (hb > 0) {                                                 // sendR[q] = my row q (the last rank: row q - 1, circularly)
            const cplx *src = X + hb + (w - hb);
            if (me == W - 1) {
                piece(src, sendR + hb, R1 - 1, hb, out_rs_, hb);
                piece(src + (size_t)(R1 - 1) * out_rs_, sendR, 1, hb, out_rs_, hb);
            } else
                piece(src, sendR, R1, hb, out_rs_, hb);
        }
        if (ha > 0) {                                                 // sendL[q] = my row q (rank 0: row q + 1, circularly)
            const cplx *src = X + hb;
            if (me == 0) {
                piece(src + out_rs_, sendL, R1 - 1, ha, out_rs_, ha);
                piece(src, sendL + (size_t)(R1 - 1) * ha, 1, ha, out_rs_, ha);
            } else
                piece(src, sendL, R1, ha, out_rs_, ha);
        }
        nph_pack = (int)ph.size();
        if (hb > 0) piece(recvL, X, R1, hb, hb, out_rs_);
        if (ha > 0) piece(recvR, X + hb + w, R1, ha, ha, out_rs_);
        const size_t nbR = (size_t)R1 * hb * sizeof(cplx), nbL = (size_t)R1 * ha * sizeof(cplx);
        if (W == 1) {                                                 // (a rank's message to itself: send and receive in one entry)
            if (nbR) xh.push_back(wfx_xfer{me, sendR, nbR, recvL, nbR});
            if (nbL) xh.push_back(wfx_xfer{me, sendL, nbL, recvR, nbL});
        } else {
            // send-only and receive-only entries: the k-th send to a peer meets the k-th receive posted for it, and with two ranks
            // both neighbours are the same peer -- "to the right" goes first on every rank, so "from the left" is received first
            if (nbR) xh.push_back(wfx_xfer{right, sendR, nbR, nullptr, 0});
            if (nbL) xh.push_back(wfx_xfer{left, sendL, nbL, nullptr, 0});
            if (nbR) xh.push_back(wfx_xfer{left, nullptr, 0, recvL, nbR});
            if (nbL) xh.push_back(wfx_xfer{right, nullptr, 0, recvR, nbL});
        }
    }
    // E4: [R1][w] rows -> own rows [nr][M] with a halo of hb points before and ha points after (circular)
    if (rows_out && !cols_) {
        cplx *X = pass_last_dst;                                     // output of the last inverse pass
        size_t off = 0;
        auto overlap = [&](int d, long long lo, long long hi, long long &a, long long &b) {
            a = std::max(cols[d], lo);
            b = std::min(cols[d + 1], hi);
            return b > a;
        };
        for (int s = 0; s < W; ++s) {
            const long long ws = cols[s + 1] - cols[s];
            cplx *src = s == me ? X + (size_t)g.rows[me] * w : (cplx *)recv + off;
            if (W > 1) {
                wfx_dist_piece p{};
                p.src = (unsigned long long)src;
                p.dst = (unsigned long long)(rows_out + hb + cols[s]);
                p.rows = nr;
                p.cols = (int)ws;
                p.src_rs = ws;
                p.dst_rs = M;
                p4.push_back(p);
                wfx_xfer x{};
                x.peer = s;
                x.send = X + (size_t)g.rows[s] * w;                  // rank s's rows of my columns (contiguous)
                x.send_bytes = (size_t)g.nrows(s) * w * sizeof(cplx);
                x.recv = src;
                x.recv_bytes = (size_t)nr * ws * sizeof(cplx);
                if (s == me) x.send = x.recv;
                x4.push_back(x);
            }
            if (s != me) off += (size_t)nr * ws;
            long long a, b;
            // halo before: row rows[dst] - 1, columns [M - hb, M)
            {
                wfx_xfer h{};
                h.peer = s;
                if (hb > 0 && overlap(me, M - hb, M, a, b)) {        // I hold part of what rank s needs
                    const int row = (g.rows[s] - 1 + R1) % R1;
                    h.send = X + (size_t)row * w + (a - cols[me]);
                    h.send_bytes = (size_t)(b - a) * sizeof(cplx);
                }
                if (hb > 0 && overlap(s, M - hb, M, a, b)) {         // rank s holds part of what I need
                    h.recv = rows_out + (a - (M - hb));
                    h.recv_bytes = (size_t)(b - a) * sizeof(cplx);
                }
                if (h.send_bytes || h.recv_bytes) x4.push_back(h);
            }
            // halo after: row rows[dst + 1], columns [0, ha)
            {
                wfx_xfer h{};
                h.peer = s;
                if (ha > 0 && overlap(me, 0, ha, a, b)) {
                    const int row = g.rows[s + 1] % R1;
                    h.send = X + (size_t)row * w + (a - cols[me]);
                    h.send_bytes = (size_t)(b - a) * sizeof(cplx);
                }
                if (ha > 0 && overlap(s, 0, ha, a, b)) {
                    h.recv = rows_out + hb + (size_t)nr * M + a;
                    h.recv_bytes = (size_t)(b - a) * sizeof(cplx);
                }
                if (h.send_bytes || h.recv_bytes) x4.push_back(h);
            }
        }
    }
    last_rows_in = rows_in;
    last_rows_out = rows_out;
}

int wfx_dist::upload_pieces()
{
    const size_t n = p1.size() + p3.size() + p4.size() + ph.size();
    oq = (n * sizeof(wfx_dist_piece) + 255) / 256 * 256;
    WFX_TRY(wfx_reserve(ctx, b_desc, oq + qmap.size() * sizeof(mr_qmap) + 256));
    std::vector<wfx_dist_piece> all;
    o1 = 0;
    all.insert(all.end(), p1.begin(), p1.end());
    o3 = all.size();
    all.insert(all.end(), p3.begin(), p3.end());
    o4 = all.size();
    all.insert(all.end(), p4.begin(), p4.end());
    oh = all.size();
    all.insert(all.end(), ph.begin(), ph.end());
    WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));               // kernels of an earlier run may still read the old descriptors
    if (n) WFX_HIP(ctx, hipMemcpy(b_desc.p, all.data(), n * sizeof(wfx_dist_piece), hipMemcpyHostToDevice));
    WFX_HIP(ctx, hipMemcpy((char *)b_desc.p + oq, qmap.data(), qmap.size() * sizeof(mr_qmap), hipMemcpyHostToDevice));
    d_first.qmap = (const mr_qmap *)((const char *)b_desc.p + oq);
    return 0;
}

// rows_in: this rank's rows of the forward transform's input; rows_out: where the inverse delivers [hb + nr M + ha] points;
// inv_in: the slab buffer (0 / 1) the inverse passes start from.  Either pointer may be null when that direction is not used.
int wfx_dist::bind_cols(const void *cols_in, long long in_rs, cplx *cols_out, long long out_rs, int inv_in)
{
    if (cols_out && out_rs < (long long)hb + w + ha) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "distributed transform: output rows of %lld points do not hold %d + %d + %d", out_rs, hb, w, ha);
    if (cols_in && in_rs < w) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "distributed transform: input rows of %lld elements do not hold %d columns", in_rs, w);
    cols_ = true;
    in_rs_ = in_rs;
    out_rs_ = out_rs;
    cols_out_ = cols_out;
    d_first.in_rs = cols_in ? in_rs : 0;
    d_last.out_rs = cols_out ? out_rs : 0;
    return bind(cols_in, cols_out, inv_in);
}

int wfx_dist::bind(const void *rows_in, cplx *rows_out, int inv_in)
{
    // the inverse slab passes ping-pong between the two slab buffers: where they end is where E3 sends from
    const int ns = d_inv.empty() ? 0 : (int)d_inv[0].size();
    const int end = (inv_in + ns) & 1;
    inv_start = inv_in ? (cplx *)b_a2.p : (cplx *)b_a.p;
    inv_result = end ? (cplx *)b_a2.p : (cplx *)b_a.p;
    build_lists(rows_in, rows_out);
    return dry_ ? 0 : upload_pieces();
}

int wfx_dist::fwd_pack_exchange(wfx_comm *c, const void *rows_in)
{
    if (rows_in != last_rows_in || !b_desc.p) return wfx_fail(ctx, WFX_ERR_STATE, "distributed transform: input rows not bound");
    const wfx_dist_piece *dp = (const wfx_dist_piece *)b_desc.p;
    long long mx = 0;
    for (const wfx_dist_piece &p : p1) mx = std::max(mx, (long long)p.rows * p.cols);
    WFX_TRY(wfx_dist_copy2d(ctx, dp + o1, (int)p1.size(), mx, es_in));
    // rows nobody owns are zero padding: the first pass reads them from the receive buffer, which other exchanges have used since
    if (g.world > 1 && g.rows_used < g.R1)
        WFX_HIP(ctx, hipMemsetAsync((char *)b_recv.p + (size_t)g.rows_used * w * es_in, 0, (size_t)(g.R1 - g.rows_used) * w * es_in, ctx->stream));
    label(c, "E1");
    return wfx_comm_exchange(c, ctx, x1.data(), (int)x1.size());
}

int wfx_dist::fwd_pass1(int in_mode)
{
    const cplx *tb = (const cplx *)tables.p;
    // P = 1: no twiddle is read; the outputs go straight to where E2 sends them from (d_first.qmap), b_y is not written
    return wfx_mr_launch_pair(ctx, d_first, tb, in_mode, 0, 0, pass1_src, (cplx *)b_y.p);
}

int wfx_dist::e2_exchange(wfx_comm *c, int chunk, int slot)
{
    label(c, "E2");
    return wfx_comm_exchange_async(c, ctx, x2c[chunk].data(), (int)x2c[chunk].size(), slot);
}

int wfx_dist::fwd_pass1_exchange(wfx_comm *c, int in_mode)
{
    WFX_TRY(fwd_pass1(in_mode));
    for (int k = 0; k < C; ++k) {
        label(c, "E2");
        WFX_TRY(wfx_comm_exchange(c, ctx, x2c[k].data(), (int)x2c[k].size()));
    }
    return 0;
}

int wfx_dist::fwd_slab_chunk(int chunk, int hilbert_spectrum, cplx **spectrum, long long skip_lo, long long skip_hi, const cplx *gtab)
{
    const cplx *tb = (const cplx *)tables.p;
    cplx *src = (cplx *)b_a.p + soff[chunk], *dst = (cplx *)b_a2.p + soff[chunk];
    const int ns = (int)d_fwd[chunk].size();
    for (int i = 0; i < ns; ++i) {
        mr_pass_desc d = d_fwd[chunk][i];
        if (i == ns - 1 && !hilbert_spectrum && skip_hi > skip_lo + 1) {
            d.skip_lo = skip_lo;
            d.skip_hi = skip_hi;
        }
        const bool skipping = d.skip_hi != 0;
        int out_mode = (hilbert_spectrum && i == ns - 1) ? 1 : ((skipping && d.ra > 0) ? 3 : 0);
        if (gtab && i == ns - 1) {       // a zero-padded convolution: this subset's slab of the transformed kernel (indexed like the slab itself)
            if (d.ra <= 0) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "distributed padded convolution needs radix-pair passes");
            d.gtab = (const double2 *)gtab;
            out_mode = 4;
        }
        WFX_TRY(wfx_mr_launch(ctx, d, tb + tw_fwd[i], 0, out_mode, 0, src, dst));
        std::swap(src, dst);
    }
    *spectrum = src;
    return 0;
}

int wfx_dist::fwd_slab(int hilbert_spectrum, cplx **spectrum, long long skip_lo, long long skip_hi, const cplx *gtab)
{
    if (C > 1) {
        if (gtab) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "distributed padded convolution: one k1 subset only");
        cplx *first = nullptr;
        for (int k = 0; k < C; ++k) {
            cplx *sp = nullptr;
            WFX_TRY(fwd_slab_chunk(k, hilbert_spectrum, &sp, skip_lo, skip_hi));
            if (k == 0) first = sp;
        }
        *spectrum = first;                 // (the subsets' results lie behind one another, as the slab buffer's layout has them)
        return 0;
    }
    const cplx *tb = (const cplx *)tables.p;
    cplx *src = (cplx *)b_a.p, *dst = (cplx *)b_a2.p;
    const int ns = (int)d_fwd[0].size();
    for (int i = 0; i < ns; ++i) {
        mr_pass_desc d = d_fwd[0][i];
        if (i == ns - 1 && !hilbert_spectrum && skip_hi > skip_lo + 1) {
            d.skip_lo = skip_lo;
            d.skip_hi = skip_hi;
        }
        const bool skipping = d.skip_hi != 0;
        int out_mode = (hilbert_spectrum && i == ns - 1) ? 1 : ((skipping && d.ra > 0) ? 3 : 0);
        if (gtab && i == ns - 1) {
            if (d.ra <= 0) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "distributed padded convolution needs radix-pair passes");
            d.gtab = (const double2 *)gtab;
            out_mode = 4;
        }
        WFX_TRY(wfx_mr_launch(ctx, d, tb + tw_fwd[i], 0, out_mode, 0, src, dst));
        std::swap(src, dst);
    }
    *spectrum = src;
    return 0;
}

int wfx_dist::inv_slab_chunk(int chunk, cplx *slab_in)
{
    const cplx *tb = (const cplx *)tables.p;
    if (slab_in != inv_start + soff[chunk] || !b_desc.p) return wfx_fail(ctx, WFX_ERR_STATE, "distributed transform: inverse input is not the bound slab buffer");
    cplx *other = (inv_start == (cplx *)b_a.p ? (cplx *)b_a2.p : (cplx *)b_a.p) + soff[chunk];
    cplx *src = slab_in, *dst = other;
    const int ns = (int)d_inv[chunk].size();
    for (int i = 0; i < ns; ++i) {
        WFX_TRY(wfx_mr_launch(ctx, d_inv[chunk][i], tb + tw_inv[i], 0, 0, 1, src, dst));
        std::swap(src, dst);
    }
    return 0;
}

int wfx_dist::e3_exchange(wfx_comm *c, int chunk, int slot)
{
    label(c, "E3");
    return wfx_comm_exchange_async(c, ctx, x3c[chunk].data(), (int)x3c[chunk].size(), slot);
}

int wfx_dist::inv_slab_exchange(wfx_comm *c, cplx *slab_in)
{
    if (slab_in != inv_start || !b_desc.p) return wfx_fail(ctx, WFX_ERR_STATE, "distributed transform: inverse input is not the bound slab buffer");
    for (int k = 0; k < C; ++k) {
        WFX_TRY(inv_slab_chunk(k, slab_in + soff[k]));
        label(c, "E3");
        WFX_TRY(wfx_comm_exchange(c, ctx, x3c[k].data(), (int)x3c[k].size()));
    }
    return 0;
}

int wfx_dist::inv_pass1_exchange(wfx_comm *c, cplx *rows_out)
{
    if (rows_out != last_rows_out || !b_desc.p) return wfx_fail(ctx, WFX_ERR_STATE, "distributed transform: output rows not bound");
    const wfx_dist_piece *dp = (const wfx_dist_piece *)b_desc.p;
    WFX_TRY(wfx_dist_scatter_k(ctx, dp + o3, (int)p3.size(), w));
    const cplx *tb = (const cplx *)tables.p;
    WFX_TRY(wfx_mr_launch_pair(ctx, d_last, tb + tw_last, 0, 0, 1, b_y.p, pass_last_dst));
    label(c, "E4");
    return wfx_comm_exchange(c, ctx, x4.data(), (int)x4.size());
}

int wfx_dist::inv_pass1_halo_exchange(wfx_comm *c)
{
    if (!cols_ || !cols_out_ || cols_out_ != last_rows_out || !b_desc.p) return wfx_fail(ctx, WFX_ERR_STATE, "distributed transform: output columns not bound");
    const wfx_dist_piece *dp = (const wfx_dist_piece *)b_desc.p;
    WFX_TRY(wfx_dist_scatter_k(ctx, dp + o3, (int)p3.size(), w));
    const cplx *tb = (const cplx *)tables.p;
    WFX_TRY(wfx_mr_launch_pair(ctx, d_last, tb + tw_last, 0, 0, 1, b_y.p, pass_last_dst));
    long long mx = 0;
    for (int i = 0; i < nph_pack; ++i) mx = std::max(mx, (long long)ph[i].rows * ph[i].cols);
    WFX_TRY(wfx_dist_copy2d(ctx, dp + oh, nph_pack, mx, 16));
    label(c, "halo");
    return wfx_comm_exchange(c, ctx, xh.data(), (int)xh.size());
}

int wfx_dist::inv_halo_unpack()
{
    if (!cols_ || !b_desc.p) return wfx_fail(ctx, WFX_ERR_STATE, "distributed transform: output columns not bound");
    const wfx_dist_piece *dp = (const wfx_dist_piece *)b_desc.p;
    long long mx = 0;
    for (size_t i = nph_pack; i < ph.size(); ++i) mx = std::max(mx, (long long)ph[i].rows * ph[i].cols);
    return wfx_dist_copy2d(ctx, dp + oh + nph_pack, (int)ph.size() - nph_pack, mx, 16);
}

unsigned long long wfx_dist::wire_bytes(int e) const
{
    unsigned long long t = 0;
    for (const wfx_xfer &x : xfers(e))
        if (x.peer != g.rank) t += x.send_bytes;
    return t;
}

int wfx_dist::inv_unpack(cplx *rows_out)
{
    if (rows_out != last_rows_out) return wfx_fail(ctx, WFX_ERR_STATE, "distributed transform: output buffer changed between exchange and unpack");
    const wfx_dist_piece *dp = (const wfx_dist_piece *)b_desc.p;
    long long mx = 0;
    for (const wfx_dist_piece &p : p4) mx = std::max(mx, (long long)p.rows * p.cols);
    return wfx_dist_copy2d(ctx, dp + o4, (int)p4.size(), mx, 16);
}

// ---- scipy.signal.resample's spectral step on slabs (the single-GPU form is resample_mr_glue in wfx_mrfft.hip) -------
// Z: forward spectrum of the packed input, slab layout [M1 / R1][B]; W: input of the inverse, slab layout [K / R1][B].
// The k1 sets are closed under negation, so the four bins an output needs (k, M1 - k, K - k, M1 - K + k) are all local.
__global__ void __launch_bounds__(256) dist_resample_glue_kernel(const cplx *__restrict__ Z, long long n0, long long num, int R1, wfx_dist_kmap km,
                                                                  cplx *__restrict__ W)
{
    const long long M1 = n0 / 2, K = num / 2, nmin = n0 < num ? n0 : num, half = nmin / 2;
    const double edge = (nmin % 2 == 0) ? (num < n0 ? 2.0 : (num > n0 ? 0.5 : 1.0)) : 1.0;
    const double inv_n0 = 1.0 / (double)n0;
    const int B = km.B;
    auto zat = [&](long long j) {                       // Z[j], j in [0, M1): all indices fit 32 bits (n0 < 2^31)
        const unsigned q = (unsigned)j / (unsigned)R1;
        const int k1 = (int)((unsigned)j - q * (unsigned)R1);
        const int kk = (k1 >= km.kb0 && k1 < km.kb0 + km.kc0) ? k1 - km.kb0 : km.kc0 + (k1 - km.kb1);
        return Z[(long long)q * B + kk];
    };
    auto bin = [&](long long j) {                       // Y[j], j in [0, K]
        if (j > half) return make_double2(0.0, 0.0);
        const long long a = j >= M1 ? j - M1 : j, b = (j == 0 || j >= M1) ? 0 : M1 - j;      // j % M1, (M1 - j) % M1 for j <= M1
        const cplx zk = zat(a), zb = zat(b);
        const cplx zc = make_double2(zb.x, -zb.y);
        double sn, cs;
        sincospi(2.0 * (double)j / (double)n0, &sn, &cs);
        const cplx dif = make_double2(zk.x - zc.x, zk.y - zc.y);
        cplx y = make_double2(0.5 * (zk.x + zc.x) + 0.5 * (cs * dif.y - sn * dif.x), 0.5 * (zk.y + zc.y) - 0.5 * (cs * dif.x + sn * dif.y));
        if (j == half) {
            y.x *= edge;
            y.y *= edge;
        }
        if (j == 0 || j == K) y.y = 0.0;
        return y;
    };
    // W[k] and W[K - k] are made of the same two bins and both live on this rank (the k1 sets are closed under negation): the lane
    // whose k is the smaller of the pair makes both, the other one has nothing to do
    auto slot = [&](long long k) {                      // position of output k in this rank's slab [K / R1][B]
        const unsigned q = (unsigned)k / (unsigned)R1;
        const int k1 = (int)((unsigned)k - q * (unsigned)R1);
        const int kk = (k1 >= km.kb0 && k1 < km.kb0 + km.kc0) ? k1 - km.kb0 : km.kc0 + (k1 - km.kb1);
        return (long long)q * B + kk;
    };
    auto emit = [&](long long e, long long k, const cplx &yk, const cplx &yr) {
        const cplx yc = make_double2(yr.x, -yr.y);
        const cplx sum = make_double2(yk.x + yc.x, yk.y + yc.y), dif = make_double2(yk.x - yc.x, yk.y - yc.y);
        double sn, cs;
        sincospi(2.0 * (double)k / (double)num, &sn, &cs);
        W[e] = make_double2((sum.x - (sn * dif.x + cs * dif.y)) * inv_n0, (sum.y + (cs * dif.x - sn * dif.y)) * inv_n0);
    };
    const long long total = (K / R1) * B;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long k2 = (long long)((unsigned)e / (unsigned)B);              // (total < 2^31)
        const int kk = (int)(e - k2 * B);
        const long long k = (long long)(kk < km.kc0 ? km.kb0 + kk : km.kb1 + (kk - km.kc0)) + k2 * R1;
        if (k != 0 && 2 * k > K) continue;               // the partner K - k < k makes this one
        const cplx yk = bin(k);
        const cplx yr = bin(K - k);
        emit(e, k, yk, yr);
        if (k != 0 && K - k != k) emit(slot(K - k), K - k, yr, yk);
    }
}

// ---- odd capture lengths: real samples against scipy's real kernel on PACKED transforms (the one-GPU form: wfx_mrfft.hip,
// wfx_dev_hilbert_conv_mr_real), here on a rank's slab [Mh / R1][B] of an Mh-point transform.  Bin k pairs with bin Mh - k, which
// lives on the same rank (k1 sets closed under negation), like the resampler's glue above.
__device__ __forceinline__ long long dist_slot(long long k, int R1, const wfx_dist_kmap &km)
{
    const unsigned q = (unsigned)k / (unsigned)R1;
    const int k1 = (int)((unsigned)k - q * (unsigned)R1);
    const int kk = (k1 >= km.kb0 && k1 < km.kb0 + km.kc0) ? k1 - km.kb0 : km.kc0 + (k1 - km.kb1);
    return (long long)q * km.B + kk;
}

// c[e] = Im G[k(e)] for every slab entry, from the packed transform Zg of the kernel's pairs (g[2q], g[2q+1]) (mr_real_kernel_untangle)
__global__ void __launch_bounds__(256) dist_real_untangle_kernel(const cplx *__restrict__ Zg, long long Mh, int R1, wfx_dist_kmap km, double *__restrict__ c)
{
    const int B = km.B;
    const long long total = (Mh / R1) * B;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long k2 = (long long)((unsigned)e / (unsigned)B);
        const int kk = (int)(e - k2 * B);
        const long long k = (long long)(kk < km.kc0 ? km.kb0 + kk : km.kb1 + (kk - km.kc0)) + k2 * R1;
        const cplx zk = Zg[e], zr = Zg[k == 0 ? e : dist_slot(Mh - k, R1, km)];
        const cplx zc = make_double2(zr.x, -zr.y);
        double sn, cs;
        sincospi((double)k / (double)Mh, &sn, &cs);
        const cplx dif = make_double2(zk.x - zc.x, zk.y - zc.y);
        c[e] = 0.5 * (zk.y + zc.y) - 0.5 * (cs * dif.x + sn * dif.y);
    }
}

// Z (the transform of the packed samples, first pass IN_MODE 1) -> W (the input of the packed inverse), in place: mr_real_conv_glue
__global__ void __launch_bounds__(256) dist_real_conv_glue_kernel(cplx *__restrict__ Z, long long Mh, int R1, wfx_dist_kmap km, const double *__restrict__ c)
{
    const int B = km.B;
    const long long total = (Mh / R1) * B;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long k2 = (long long)((unsigned)e / (unsigned)B);
        const int kk = (int)(e - k2 * B);
        const long long k = (long long)(kk < km.kc0 ? km.kb0 + kk : km.kb1 + (kk - km.kc0)) + k2 * R1;
        if (k == 0) {                                                    // Y[0] = Y[Mh] = 0: the kernel sums to zero both ways
            Z[e] = make_double2(0.0, 0.0);
            continue;
        }
        const long long r = Mh - k;
        if (k > r) continue;                                             // the partner makes this one
        const long long er = dist_slot(r, R1, km);
        const cplx a = Z[e], b = Z[er];                                  // Z'[k], Z'[Mh - k]
        const cplx zk = make_double2(b.y, b.x), zm = make_double2(a.y, a.x);      // Z[k] = i conj Z'[r], Z[r] = i conj Z'[k]
        double sn, cs;
        sincospi((double)k / (double)Mh, &sn, &cs);
        const cplx zmc = make_double2(zm.x, -zm.y), zkc = make_double2(zk.x, -zk.y);
        const cplx dk = make_double2(zk.x - zmc.x, zk.y - zmc.y), dr = make_double2(zm.x - zkc.x, zm.y - zkc.y);
        const cplx xk = make_double2(0.5 * (zk.x + zmc.x) + 0.5 * (cs * dk.y - sn * dk.x), 0.5 * (zk.y + zmc.y) - 0.5 * (cs * dk.x + sn * dk.y));
        const cplx xr = make_double2(0.5 * (zm.x + zkc.x) + 0.5 * (-cs * dr.y - sn * dr.x), 0.5 * (zm.y + zkc.y) - 0.5 * (-cs * dr.x + sn * dr.y));
        const double ck = c[e], cr = c[er];
        const cplx yk = make_double2(-xk.y * ck, xk.x * ck), yr = make_double2(-xr.y * cr, xr.x * cr);
        const cplx yrc = make_double2(yr.x, -yr.y), ykc = make_double2(yk.x, -yk.y);
        const cplx sk = make_double2(yk.x + yrc.x, yk.y + yrc.y), ek = make_double2(yk.x - yrc.x, yk.y - yrc.y);
        const cplx sr = make_double2(yr.x + ykc.x, yr.y + ykc.y), er2 = make_double2(yr.x - ykc.x, yr.y - ykc.y);
        Z[e] = make_double2(sk.x - (sn * ek.x + cs * ek.y), sk.y + (cs * ek.x - sn * ek.y));
        if (r != k) Z[er] = make_double2(sr.x - (sn * er2.x - cs * er2.y), sr.y - (cs * er2.x + sn * er2.y));
    }
}

int wfx_dist_real_untangle_km(wfx_ctx *ctx, int R1, const wfx_dist_kmap &km, const cplx *Zg, long long Mh, double *ctab)
{
    const long long total = (Mh / R1) * km.B;
    WFX_LAUNCH(ctx, K_RESAMPLE_PW, dist_real_untangle_kernel, dim3(wfx_stream_grid((uint64_t)total, 256)), dim3(256), Zg, Mh, R1, km, ctab);
    return 0;
}

int wfx_dist_real_conv_glue_km(wfx_ctx *ctx, int R1, const wfx_dist_kmap &km, cplx *Z, long long Mh, const double *ctab)
{
    const long long total = (Mh / R1) * km.B;
    WFX_LAUNCH(ctx, K_RESAMPLE_PW, dist_real_conv_glue_kernel, dim3(wfx_stream_grid((uint64_t)total, 256)), dim3(256), Z, Mh, R1, km, ctab);
    return 0;
}

int wfx_dist_real_untangle(wfx_ctx *ctx, const wfx_dist_geom &g, const cplx *Zg, long long Mh, double *ctab)
{
    return wfx_dist_real_untangle_km(ctx, g.R1, g.km[g.rank], Zg, Mh, ctab);
}

int wfx_dist_real_conv_glue(wfx_ctx *ctx, const wfx_dist_geom &g, cplx *Z, long long Mh, const double *ctab)
{
    return wfx_dist_real_conv_glue_km(ctx, g.R1, g.km[g.rank], Z, Mh, ctab);
}

int wfx_dist_resample_glue_km(wfx_ctx *ctx, int R1, const wfx_dist_kmap &km, const cplx *Z, long long n0, long long num, cplx *W)
{
    const long long total = (num / 2 / R1) * km.B;
    WFX_LAUNCH(ctx, K_RESAMPLE_PW, dist_resample_glue_kernel, dim3(wfx_stream_grid((uint64_t)total, 256)), dim3(256), Z, n0, num, R1, km, W);
    return 0;
}

int wfx_dist_resample_glue(wfx_ctx *ctx, const wfx_dist_geom &g, const cplx *Z, long long n0, long long num, cplx *W)
{
    return wfx_dist_resample_glue_km(ctx, g.R1, g.km[g.rank], Z, n0, num, W);
}
