// Distributed form of the two global operators of the path (wfx_dist.hip), used by the sharded decode (wfx_shard.hip).
//
// A transform of L = R1 * M points (R1 = first radix of the plan, a radix pair with a register-resident pass) over
// `world` ranks.  Index n = n1 M + n2.  Three layouts of one rank's share:
//
//   rows     n1 in [rows[r], rows[r+1]), all n2                 -- contiguous sample ranges: what the stencil stages own
//   columns  all n1, n2 in [cols[r], cols[r+1])                 -- [R1][w]: the first pass (R1-point transforms along n1)
//   slab     first-pass outputs k1 in K_r (a set closed under k1 -> R1 - k1), all n2 / all k2
//            -- [M][B] with the k1 index innermost: the remaining passes are B interleaved M-point transforms
//
// forward:  rows -(E1)-> columns, pass 1, -(E2)-> slab, passes 2..np            (spectrum: X[k1 + R1 k2] at [k2][kk])
// inverse:  slab, passes np..2, -(E3)-> columns, pass 1, -(E4)-> rows (+ a halo of neighbouring points)
//
// Every exchange E is one personalised all-to-all (grouped send / recv on the library's stream) between a packing and an
// unpacking copy; the passes themselves are the single-GPU kernels (mr2_pass) with a generalised twiddle index.
#pragma once

#include "wfx_internal.h"

struct wfx_dist_kmap {       // k1 = kk < kc0 ? kb0 + kk : kb1 + (kk - kc0), kk in [0, B)
    int kb0, kc0, kb1, B;
};

struct wfx_dist_geom {       // what all transforms of one sharded decode share
    int world = 1, rank = 0;
    int ra1 = 0, rb1 = 0, R1 = 0;
    // rows dealt to the ranks: all R1 of them, or -- a zero-padded convolution's INPUT and OUTPUT -- only the first rows_used: the
    // rows behind them are zero padding going in (nobody sends them: the receiver clears its copy) and garbage coming out (nobody
    // wants them), so E1 and E4 carry rows_used / R1 of their bytes and every rank owns an equal share of the real samples
    int rows_used = 0;
    std::vector<int> rows;               // [world + 1], rows[world] = rows_used
    std::vector<wfx_dist_kmap> km;       // [world]
    int nrows(int r) const { return rows[r + 1] - rows[r]; }
};

// host-only: partitions for a first radix (ra1, rb1); false when the world is too large for it
bool wfx_dist_make_geom(wfx_dist_geom &g, int world, int rank, int ra1, int rb1, int rows_used = 0);
// host-only: first radix for transforms of the given lengths (all must be multiples of it with pair-decomposable cofactors)
bool wfx_dist_choose_r1(const long long *lengths, int nlen, int world, int *ra1, int *rb1, bool pairs_required = false);

struct wfx_dist_piece {      // one 2-D block of a packing / unpacking copy (device descriptor)
    unsigned long long src, dst;         // element-typed base addresses
    int rows, cols;
    long long src_rs, dst_rs;
    int kb0, kc0, kb1, B;                // column map (gather / scatter by k1)
};

class wfx_dist {
  public:
    // buffers are owned by this object; `halo_before` / `halo_after`: points delivered around the own rows by the inverse
    // dry: plan only -- no device memory, no kernels; buffers get fake base addresses `dry_base + k * 2^36` so that exchange
    // lists and piece descriptors can be built and checked on a machine without a GPU (wfx_shard_dry_run)
    // nchunk > 1 (columns mode): the rank's first-pass outputs k1 are cut into that many subsets -- each closed under k1 -> R1 - k1
    // like the rank's whole set, i.e. the k1 sets of a geometry with world * nchunk ranks -- and E2, the slab passes and E3 run
    // subset by subset, so that the exchange of one subset (on the communicator's own stream) overlaps the passes of another
    int init(wfx_ctx *ctx, const wfx_dist_geom &g, long long L, int elem_bytes_in, int halo_before, int halo_after, bool dry = false,
             unsigned long long dry_base = 0, int nchunk = 1);
    int chunks() const { return C; }
    void release();
    void set_tag(const char *t) { snprintf(tag_, sizeof tag_, "%s", t); }        // names this transform's exchanges in the wire statistics
    const char *tag() const { return tag_; }
    // (for the dry run) exchange e in 1..4: its messages and its copy pieces; the buffers as (base, bytes) pairs
    const std::vector<wfx_xfer> &xfers(int e) const { return e == 1 ? x1 : e == 2 ? x2 : e == 3 ? x3 : e == 4 ? x4 : xh; }      // (E2 / E3: all subsets, in order)
    const std::vector<wfx_xfer> &xfers_chunk(int e, int c) const { return e == 2 ? x2c[c] : x3c[c]; }
    const wfx_dist_kmap &chunk_kmap(int c) const { return kmc[c]; }
    const std::vector<wfx_dist_piece> &pieces(int e) const { return e == 1 ? p1 : e == 2 ? p_none : e == 3 ? p3 : e == 4 ? p4 : ph; }
    const std::vector<mr_qmap> &first_pass_map() const { return qmap; }
    void buffers(std::vector<std::pair<unsigned long long, unsigned long long>> &out) const;
    // rows_in: this rank's rows of the forward input; rows_out: [halo_before + nr M + halo_after] points delivered by the
    // inverse; inv_in: slab buffer (0 / 1) the inverse starts from.  Call once the buffers exist, before the first run.
    int bind(const void *rows_in, cplx *rows_out, int inv_in);
    // COLUMNS mode (round 4): the stages around the transform live in the columns layout too, so E1 and E4 do not exist.
    //   cols_in : this rank's w columns of every row n1, rows `in_rs` elements (of elem_bytes_in) apart: the first pass reads them
    //             in place
    //   cols_out: rows of [halo_before | w own points | halo_after], `out_rs` points apart: the last inverse pass writes the own
    //             points of row n1 at cols_out + n1 out_rs + halo_before, inv_pass1_halo_exchange then fetches the halos from the
    //             neighbouring ranks' columns -- the points before (n1, cols[r]) are the last columns of row n1 on rank r - 1, or,
    //             on rank 0, of row n1 - 1 on the last rank (circular: both transforms are cyclic); likewise behind
    // Either pointer may be null when that direction is not used.
    int bind_cols(const void *cols_in, long long in_rs, cplx *cols_out, long long out_rs, int inv_in);
    bool cols_mode() const { return cols_; }
    // slab buffer index in which fwd_slab leaves the spectrum
    int fwd_result_index() const { return d_fwd.empty() ? 0 : (int)(d_fwd[0].size() & 1); }

    long long L = 0, M = 0;
    int w = 0, B = 0, nr = 0;            // this rank's columns, slab batch, rows
    std::vector<long long> cols;         // [world + 1]
    long long slab_points() const { return M * (long long)B; }
    // ---- subset by subset (nchunk >= 1; `slot`: which of the communicator's completion events the exchange records) -------
    int fwd_pass1(int in_mode);                                                    // pass 1: every subset's E2 messages are ready
    int e2_exchange(wfx_comm *c, int chunk, int slot);                             // E2 of one subset (asynchronous where the transport can)
    int fwd_slab_chunk(int chunk, int hilbert_spectrum, cplx **spectrum, long long skip_lo = 0, long long skip_hi = 0, const cplx *gtab = nullptr);
    long long chunk_offset(int chunk) const { return soff[chunk]; }             // points in front of the subset's slab in a slab buffer
    int inv_slab_chunk(int chunk, cplx *slab_in);                                  // passes np..2 of one subset
    int e3_exchange(wfx_comm *c, int chunk, int slot);
    cplx *slab_chunk(int i, int chunk) { return slab_buffer(i) + soff[chunk]; }
    int fwd_result_index_chunk() const { return fwd_result_index(); }

    // ---- forward --------------------------------------------------------------------------------------------------
    // rows_in: this rank's rows [nr][M] (elements of elem_bytes_in: 16 = cplx / pairs of doubles, 4 = int16 pairs)
    int fwd_pack_exchange(wfx_comm *c, const void *rows_in);                       // E1
    int fwd_pass1_exchange(wfx_comm *c, int in_mode);                              // pass 1, E2
    // passes 2..np; bins with skip_lo < k < skip_hi (global indices) are not stored by the last one (skip_hi == 0: all are)
    // gtab: the last pass multiplies output o (slab layout) by gtab[o] -- a zero-padded convolution's transformed kernel,
    // computed by these very passes (any-length sharded Hilbert transform); needs radix-pair passes
    int fwd_slab(int hilbert_spectrum, cplx **spectrum, long long skip_lo = 0, long long skip_hi = 0, const cplx *gtab = nullptr);
    // ---- inverse ---------------------------------------------------------------------------------------------------
    // slab_in: [M][B] in one of slab_buffer(0 / 1); destroyed
    int inv_slab_exchange(wfx_comm *c, cplx *slab_in);                             // passes np..2, E3
    int inv_pass1_exchange(wfx_comm *c, cplx *rows_out);                           // unpack, pass 1, E4 (halo pieces land in rows_out directly)
    int inv_unpack(cplx *rows_out);                                                // rows_out: [halo_before + nr M + halo_after]
    int inv_pass1_halo_exchange(wfx_comm *c);                                      // columns mode: unpack E3, pass 1, the halo exchange
    int inv_halo_unpack();                                                         // columns mode: received halo columns into their rows
    // bytes this rank sends to OTHER ranks in exchange e (1..4; 5 = the halo exchange of the columns mode): the wire model's input
    unsigned long long wire_bytes(int e) const;
    cplx *slab_buffer(int i) { return (cplx *)(i ? b_a2.p : b_a.p); }

  private:
    wfx_ctx *ctx = nullptr;
    wfx_dist_geom g;
    int es_in = 16, hb = 0, ha = 0;
    bool dry_ = false;
    std::vector<wfx_mr_radix> sub;                 // passes of the M-point transforms (pairs where possible)
    mr_pass_desc d_first{}, d_last{};
    std::vector<std::vector<mr_pass_desc>> d_fwd, d_inv;     // [subset][pass]
    int C = 1;                                     // subsets of this rank's k1 set
    std::vector<wfx_dist_kmap> kmc;                // [C] this rank's subsets; kmv: every rank's, [world * C]
    std::vector<wfx_dist_kmap> kmv;
    std::vector<long long> soff;                   // [C + 1] offsets of the subsets' slabs (points) in the slab buffers
    std::vector<std::vector<wfx_xfer>> x2c, x3c;   // [C]
    std::vector<size_t> tw_fwd, tw_inv;
    size_t tw_last = 0;
    wfx_devbuf tables, b_pack, b_recv, b_y, b_a, b_a2, b_desc, b_halo;
    char tag_[16] = "transform";
    void label(wfx_comm *c, const char *e) const;
    bool cols_ = false;
    long long in_rs_ = 0, out_rs_ = 0;
    std::vector<wfx_xfer> xh;                       // columns mode: the halo exchange
    std::vector<wfx_dist_piece> ph;                 // its packing pieces [0, nph_pack) and unpacking pieces behind them
    int nph_pack = 0;
    size_t oh = 0;
    cplx *cols_out_ = nullptr;
    cplx *inv_result = nullptr, *inv_start = nullptr;
    // exchange lists and piece descriptors (device copies in b_desc)
    std::vector<wfx_xfer> x1, x2, x3, x4;
    std::vector<wfx_dist_piece> p1, p3, p4;
    std::vector<mr_qmap> qmap;                      // E2: where the first pass stores each of its R1 outputs
    size_t oq = 0;                                  // byte offset of the qmap table in b_desc
    const void *pass1_src = nullptr;                // rows_in itself when there is one rank, the received columns otherwise
    cplx *pass_last_dst = nullptr;                  // likewise for the last inverse pass
    std::vector<wfx_dist_piece> p_none;
    size_t o1 = 0, o3 = 0, o4 = 0;                  // offsets of the piece arrays in b_desc
    const void *last_rows_in = nullptr;
    cplx *last_rows_out = nullptr;
    void build_lists(const void *rows_in, cplx *rows_out);
    int upload_pieces();
};

// copies driven by piece descriptors (one workgroup column per piece: grid.y)
int wfx_dist_copy2d(wfx_ctx *ctx, const wfx_dist_piece *dev_pieces, int npieces, long long max_elems, int elem_bytes);
int wfx_dist_scatter_k(wfx_ctx *ctx, const wfx_dist_piece *dev_pieces, int npieces, int max_rows);
// scipy.signal.resample's bin copy between the forward spectrum slab (packed n0 / 2 points) and the inverse's input slab (num / 2)
int wfx_dist_resample_glue(wfx_ctx *ctx, const wfx_dist_geom &g, const cplx *Z, long long n0, long long num, cplx *W);
// the same for one k1 subset of a rank (its own slab [.][km.B] in, its own slab out)
int wfx_dist_resample_glue_km(wfx_ctx *ctx, int R1, const wfx_dist_kmap &km, const cplx *Z, long long n0, long long num, cplx *W);
// odd capture lengths on packed real transforms (the sharded counterpart of wfx_dev_hilbert_conv_mr_real): the kernel's table
// c[slab entry] from the packed transform of its pairs, and the glue between the samples' forward and inverse transforms, in place
int wfx_dist_real_untangle(wfx_ctx *ctx, const wfx_dist_geom &g, const cplx *Zg, long long Mh, double *ctab);
int wfx_dist_real_conv_glue(wfx_ctx *ctx, const wfx_dist_geom &g, cplx *Z, long long Mh, const double *ctab);
// the same for one k1 subset of a rank (its own slab [.][km.B])
int wfx_dist_real_untangle_km(wfx_ctx *ctx, int R1, const wfx_dist_kmap &km, const cplx *Zg, long long Mh, double *ctab);
int wfx_dist_real_conv_glue_km(wfx_ctx *ctx, int R1, const wfx_dist_kmap &km, cplx *Z, long long Mh, const double *ctab);
