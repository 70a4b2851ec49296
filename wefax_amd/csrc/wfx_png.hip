// save_output_image (wefax.py:407-408) for the image of the decode that just ran: the PNG file is assembled ON THE DEVICE.
//
// After the kernels, the slowest step of `python wefax.py in.wav 120 out.png` was PNG encoding on the host (55 of 72 ms for the
// 10-minute capture: filtering, deflate, Adler-32 and CRC-32 of 27 MB at 1-2.5 GB/s per core).  Here the decoder's image never
// takes that route: one kernel lays out the zlib stream -- filter byte 0 + the row's pixels, in STORED deflate blocks of whole
// rows (no compression: the output is a valid PNG with the same pixels as the reference's, larger on disk) -- while computing
// the per-row sums of Adler-32; a second kernel takes the CRC-32 of the stream in 8-KiB segments; the host combines the
// partial checksums in closed form (a few thousand integer operations), copies the finished file image from pinned memory
// and writes it out.  8-bit gray, width x 4 height, no interlace -- the layout PIL's writer gives the reference.
#include <cstdio>
#include <cstring>

#include "wfx_internal.h"
#include <algorithm>
#include <atomic>
#include <mutex>
#include <chrono>
#include <cstdlib>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <unistd.h>

#define PNG_SEG 65536         // bytes per CRC span (one workgroup)

__device__ __forceinline__ unsigned png_stream_byte(unsigned long long pos, const uint8_t *__restrict__ img, unsigned w, unsigned h, unsigned rpb,
                                                    unsigned long long stream_len)
{
    // zlib header | stored blocks {BFINAL, LEN, ~LEN, rows of (filter 0, pixels)} | Adler-32 (filled in by the host)
    if (pos < 2) return pos == 0 ? 0x78u : 0x01u;
    if (pos >= stream_len - 4) return 0u;
    const unsigned long long q = pos - 2;
    const unsigned row_bytes = w + 1, full = 5u + rpb * row_bytes;
    const unsigned blk = (unsigned)(q / full), r = (unsigned)(q - (unsigned long long)blk * full);
    const unsigned first_row = blk * rpb;
    const unsigned rows = h - first_row < rpb ? h - first_row : rpb;
    if (r < 5) {
        const unsigned len = rows * row_bytes;
        switch (r) {
        case 0: return first_row + rows >= h ? 1u : 0u;
        case 1: return len & 255u;
        case 2: return (len >> 8) & 255u;
        case 3: return (~len) & 255u;
        default: return ((~len) >> 8) & 255u;
        }
    }
    const unsigned d = r - 5, rr = d / row_bytes, c = d - rr * row_bytes;
    return c == 0 ? 0u : (unsigned)img[(unsigned long long)(first_row + rr) * w + (c - 1)];
}

// four stream bytes per thread
__global__ void __launch_bounds__(256) png_pack_kernel(const uint8_t *__restrict__ img, unsigned w, unsigned h, unsigned rpb, unsigned long long stream_len,
                                                        unsigned *__restrict__ out_words)
{
    const unsigned long long nwords = (stream_len + 3) / 4;
    for (unsigned long long i = blockIdx.x * 256ull + threadIdx.x; i < nwords; i += (unsigned long long)gridDim.x * 256ull) {
        unsigned v = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned long long pos = 4 * i + k;
            if (pos < stream_len) v |= png_stream_byte(pos, img, w, h, rpb, stream_len) << (8 * k);
        }
        out_words[i] = v;
    }
}

// Adler-32 partial sums of one row of the unfiltered-stream data (filter byte 0, then the pixels): S = sum d, W = sum d (n - c), c = column
// in the row of n = w + 1 bytes.  One wave per row.
__global__ void __launch_bounds__(256) png_adler_rows_kernel(const uint8_t *__restrict__ img, unsigned w, unsigned h, unsigned long long *__restrict__ sums)
{
    const unsigned row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= h) return;
    unsigned long long s = 0, wsum = 0;
    const unsigned n = w + 1;
    for (unsigned x = lane; x < w; x += 64) {
        const unsigned d = img[(unsigned long long)row * w + x];
        s += d;
        wsum += (unsigned long long)d * (n - (x + 1));
    }
    for (int off = 32; off > 0; off >>= 1) {
        s += __shfl_down(s, off);
        wsum += __shfl_down(wsum, off);
    }
    if (lane == 0) {
        sums[2 * row] = s;
        sums[2 * row + 1] = wsum;
    }
}

// raw CRC-32 register (reflected polynomial 0xEDB88320, initial value 0, no final xor) of each PNG_SEG-byte span.  A workgroup
// takes one span: 256 lanes run the table form over 256 bytes each, then the partial registers are joined pairwise,
// crc(A || B) = Z_len(B) crc(A) ^ crc(B) with Z_n the GF(2) operator of n zero bytes (ops[level]: n = 256 << level).
// (One lane per 8 KiB segment, as this kernel began, left 47 waves walking 8192 dependent table look-ups each: 4 ms.)
__global__ void __launch_bounds__(256) png_crc_spans_kernel(const uint8_t *__restrict__ data, const unsigned *__restrict__ ops, unsigned *__restrict__ crcs)
{
    __shared__ unsigned table[256];
    __shared__ unsigned part[256];
    __shared__ unsigned op[8][32];
    const int t = threadIdx.x;
    {
        unsigned c = (unsigned)t;
        for (int k = 0; k < 8; ++k) c = (c & 1u) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
        table[t] = c;
        op[t >> 5][t & 31] = ops[t];
    }
    __syncthreads();
    const uint4 *p = (const uint4 *)(data + (size_t)blockIdx.x * PNG_SEG + (size_t)t * 256);      // spans start 16-byte aligned
    uint4 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = p[i];
    unsigned c = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const unsigned wv[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned x = wv[k];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                c = table[(c ^ x) & 255u] ^ (c >> 8);
                x >>= 8;
            }
        }
    }
    part[t] = c;
    __syncthreads();
#pragma unroll
    for (int level = 0; level < 8; ++level) {
        const int stride = 1 << level;
        if ((t & (2 * stride - 1)) == 0) {
            unsigned left = part[t], sum = 0;
            for (int i = 0; left; left >>= 1, ++i)
                if (left & 1u) sum ^= op[level][i];
            part[t] = sum ^ part[t + stride];
        }
        __syncthreads();
    }
    if (t == 0) crcs[blockIdx.x] = part[0];
}


// ---- device deflate ------------------------------------------------------------------------------------------------
// The compressed form of the same file (wfx_decode_png_ex(..., deflate = 1)): rows are "Up"-filtered (the 4x vertical
// interpolation makes consecutive rows nearly equal), the filtered stream is cut into chunks of whole rows (<= 34 KB, one
// workgroup each) and every chunk becomes ONE dynamic-Huffman deflate block followed by an empty stored block, which ends it on a
// byte boundary (zlib's sync flush, the pigz construction), so that chunks are encoded independently and concatenated.
//   tokens: literals and distance-1 matches (runs of one byte value: flat picture areas filter to runs of zeros); a lane
//           tokenises its own span of ~130 bytes, a run never crosses a span
//   code:   ONE literal/length code per image, built on the host from the histogram of all tokens (png_hist_kernel), length-
//           limited to 15 bits; the distance code has the single symbol "1"
//   png_encode_kernel: span bit counts -> exclusive scan -> every lane ORs its bits into the chunk's LDS image; a chunk that
//           would not shrink is emitted as a stored block instead
//   png_scan_kernel / png_gather_kernel: chunk sizes -> offsets -> the contiguous zlib stream
// Adler-32 (of the FILTERED stream) comes from per-row sums as in the stored form, CRC-32 from the same span kernel.
#define PNG_DEF_MAXRAW 34816
#define PNG_DEF_CAP (PNG_DEF_MAXRAW + 64)      // bytes of scratch per chunk (>= stored form + slack, multiple of 4)
#define PNG_HDR_WORDS 80

struct png_codes {
    unsigned lit[256];           // (nbits << 24) | bits, bits in stream order (Huffman code bit-reversed)
    unsigned mt[256];            // run of 3 + i more bytes at distance 1: length code + extra bits + the distance code
    unsigned eob;
    unsigned hdr_bits;           // block header: BFINAL = 0, BTYPE = 10, HLIT / HDIST / HCLEN, the code lengths
    unsigned hdr[PNG_HDR_WORDS];
};

__device__ __forceinline__ unsigned png_len_symbol(unsigned len)        // deflate length symbol - 257 for a match of `len` bytes (3..258)
{
    if (len == 258) return 28;
    const unsigned l = len - 3;
    if (l < 8) return l;
    const unsigned k = 29 - __clz(l);                                     // extra bits: floor(log2 l) - 2
    return 4 * k + 4 + ((l >> k) & 3);
}

// residuals of rows [r0, r1) into LDS: per row the filter byte 2, then pixel - pixel above (the row above row 0 is zero)
__device__ __forceinline__ void png_load_chunk(const uint8_t *__restrict__ img, unsigned w, unsigned r0, unsigned r1, uint8_t *res)
{
    const unsigned n = w + 1, t = threadIdx.x;
    const bool quads = (w & 3) == 0 && ((unsigned long long)img & 3) == 0;
    for (unsigned r = r0; r < r1; ++r) {
        uint8_t *dst = res + (r - r0) * n;
        const uint8_t *cur = img + (unsigned long long)r * w, *up = cur - w;
        if (t == 0) dst[0] = 2;
        if (quads) {
            for (unsigned x = 4 * t; x < w; x += 1024) {
                const uchar4 c = *(const uchar4 *)(cur + x);
                uchar4 u = make_uchar4(0, 0, 0, 0);
                if (r) u = *(const uchar4 *)(up + x);
                dst[1 + x] = (uint8_t)(c.x - u.x);
                dst[2 + x] = (uint8_t)(c.y - u.y);
                dst[3 + x] = (uint8_t)(c.z - u.z);
                dst[4 + x] = (uint8_t)(c.w - u.w);
            }
        } else {
            for (unsigned x = t; x < w; x += 256) dst[1 + x] = (uint8_t)(cur[x] - (r ? up[x] : 0));
        }
    }
}

// walks the tokens of [a, b): lit(value) / run(length) callbacks; the same walk in both kernels
template <typename LIT, typename RUN>
__device__ __forceinline__ void png_tokens(const uint8_t *res, unsigned a, unsigned b, LIT lit, RUN run)
{
    unsigned i = a;
    while (i < b) {
        const unsigned v = res[i];
        lit(v);
        unsigned j = i + 1;
        while (j < b && res[j] == v && j - i - 1 < 258) ++j;
        const unsigned len = j - i - 1;
        if (len >= 3) {
            run(len);
            i = j;
        } else {
            ++i;
        }
    }
}

__global__ void __launch_bounds__(256) png_hist_kernel(const uint8_t *__restrict__ img, unsigned w, unsigned h, unsigned rpc, unsigned *__restrict__ hist,
                                                        unsigned long long *__restrict__ sums)
{
    __shared__ __attribute__((aligned(16))) uint8_t res[PNG_DEF_MAXRAW];
    __shared__ unsigned hh[288];
    __shared__ unsigned long long rs[64][2];
    const unsigned t = threadIdx.x, n = w + 1;
    const unsigned r0 = blockIdx.x * rpc, r1 = min(h, r0 + rpc), raw = (r1 - r0) * n;
    for (unsigned i = t; i < 288; i += 256) hh[i] = 0;
    if (t < 64) rs[t][0] = rs[t][1] = 0;
    png_load_chunk(img, w, r0, r1, res);
    __syncthreads();
    // Adler-32 row sums of the filtered stream: S = sum d, W = sum d (n - c)
    for (unsigned r = 0; r < r1 - r0; ++r) {
        unsigned long long sa = 0, sw = 0;
        for (unsigned c = t; c < n; c += 256) {
            const unsigned d = res[r * n + c];
            sa += d;
            sw += (unsigned long long)d * (n - c);
        }
        for (int off = 32; off > 0; off >>= 1) {
            sa += __shfl_down(sa, off);
            sw += __shfl_down(sw, off);
        }
        if ((t & 63) == 0) {
            atomicAdd(&rs[r][0], sa);
            atomicAdd(&rs[r][1], sw);
        }
    }
    const unsigned span = (raw + 255) / 256;
    const unsigned a = min(raw, t * span), b = min(raw, a + span);
    png_tokens(res, a, b, [&](unsigned v) { atomicAdd(&hh[v], 1u); }, [&](unsigned len) { atomicAdd(&hh[257 + png_len_symbol(len)], 1u); });
    __syncthreads();
    for (unsigned i = t; i < 288; i += 256)
        if (hh[i]) atomicAdd(&hist[i], hh[i]);
    if (t < r1 - r0) {
        sums[2 * (r0 + t)] = rs[t][0];
        sums[2 * (r0 + t) + 1] = rs[t][1];
    }
}

__global__ void __launch_bounds__(256) png_encode_kernel(const uint8_t *__restrict__ img, unsigned w, unsigned h, unsigned rpc, const png_codes *__restrict__ codes,
                                                          uint8_t *__restrict__ scratch, unsigned *__restrict__ sizes)
{
    __shared__ __attribute__((aligned(16))) uint8_t res[PNG_DEF_MAXRAW];
    __shared__ unsigned out[PNG_DEF_CAP / 4];
    __shared__ unsigned lit[256], mt[256];
    __shared__ unsigned scan[256];
    const unsigned t = threadIdx.x, n = w + 1;
    const unsigned r0 = blockIdx.x * rpc, r1 = min(h, r0 + rpc), raw = (r1 - r0) * n;
    lit[t] = codes->lit[t];
    mt[t] = codes->mt[t];
    for (unsigned i = t; i < PNG_DEF_CAP / 4; i += 256) out[i] = 0;
    png_load_chunk(img, w, r0, r1, res);
    __syncthreads();
    const unsigned span = (raw + 255) / 256;
    const unsigned a = min(raw, t * span), b = min(raw, a + span);
    unsigned bits = 0;
    png_tokens(res, a, b, [&](unsigned v) { bits += lit[v] >> 24; }, [&](unsigned len) { bits += mt[len - 3] >> 24; });
    scan[t] = bits;
    __syncthreads();
    for (unsigned off = 1; off < 256; off <<= 1) {           // inclusive scan
        const unsigned v = t >= off ? scan[t - off] : 0;
        __syncthreads();
        scan[t] += v;
        __syncthreads();
    }
    const unsigned hdr_bits = codes->hdr_bits, eob = codes->eob;
    const unsigned body_bits = scan[255], start = hdr_bits + scan[t] - bits;
    const unsigned end_bits = hdr_bits + body_bits + (eob >> 24) + 3;         // ... + the empty stored block's three header bits
    const unsigned nbytes = (end_bits + 7) / 8 + 4;                             // + LEN = 0, NLEN = 0xffff
    uint8_t *dst = scratch + (size_t)blockIdx.x * PNG_DEF_CAP;
    if (nbytes > raw + 5) {
        // would not shrink: one stored block (the chunk starts on a byte boundary)
        if (t == 0) {
            dst[0] = 0;
            dst[1] = (uint8_t)(raw & 255);
            dst[2] = (uint8_t)(raw >> 8);
            dst[3] = (uint8_t)(~raw & 255);
            dst[4] = (uint8_t)((~raw >> 8) & 255);
            sizes[blockIdx.x] = raw + 5;
        }
        for (unsigned i = t; i < raw; i += 256) dst[5 + i] = res[i];
        return;
    }
    for (unsigned i = t; i < (hdr_bits + 31) / 32; i += 256) atomicOr(&out[i], codes->hdr[i]);
    {
        // bit writer: `acc` holds the bits not yet stored, the low `fill` of them; the first and the last word of a lane are shared
        // with its neighbours (atomic OR into the zeroed image), the words between are its own
        unsigned long long acc = 0;
        unsigned fill = start & 31, word = start >> 5;
        bool first = true;
        auto put = [&](unsigned code) {
            acc |= (unsigned long long)(code & 0xffffffu) << fill;
            fill += code >> 24;
            if (fill >= 32) {
                if (first)
                    atomicOr(&out[word], (unsigned)acc);
                else
                    out[word] = (unsigned)acc;
                first = false;
                ++word;
                acc >>= 32;
                fill -= 32;
            }
        };
        png_tokens(res, a, b, [&](unsigned v) { put(lit[v]); }, [&](unsigned len) { put(mt[len - 3]); });
        if (fill) atomicOr(&out[word], (unsigned)acc);
    }
    __syncthreads();
    if (t == 0) {
        const unsigned pos = hdr_bits + body_bits;
        const unsigned long long e = (unsigned long long)(eob & 0xffffffu) << (pos & 31);
        atomicOr(&out[pos >> 5], (unsigned)e);
        if (e >> 32) atomicOr(&out[(pos >> 5) + 1], (unsigned)(e >> 32));
        // 000 (BFINAL = 0, BTYPE = 00), zero padding to the byte, LEN = 0x0000, NLEN = 0xffff
        const unsigned byte0 = (end_bits + 7) / 8 + 2;
        uint8_t *ob = (uint8_t *)out;
        ob[byte0] = 0xff;
        ob[byte0 + 1] = 0xff;
        sizes[blockIdx.x] = nbytes;
    }
    __syncthreads();
    for (unsigned i = t; i < (nbytes + 3) / 4; i += 256) ((unsigned *)dst)[i] = out[i];
}

// offsets[c] = 2 + sizes[0] + ... + sizes[c - 1] (after the zlib header); meta[0] = stream length incl. final block and Adler-32
__global__ void __launch_bounds__(256) png_scan_kernel(const unsigned *__restrict__ sizes, unsigned nchunks, unsigned long long *__restrict__ offsets,
                                                        unsigned long long *__restrict__ meta)
{
    __shared__ unsigned long long part[256];
    const unsigned t = threadIdx.x;
    const unsigned per = (nchunks + 255) / 256;
    const unsigned a = min(nchunks, t * per), b = min(nchunks, a + per);
    unsigned long long s = 0;
    for (unsigned i = a; i < b; ++i) s += sizes[i];
    part[t] = s;
    __syncthreads();
    for (unsigned off = 1; off < 256; off <<= 1) {
        const unsigned long long v = t >= off ? part[t - off] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    unsigned long long run = 2 + part[t] - s;
    for (unsigned i = a; i < b; ++i) {
        offsets[i] = run;
        run += sizes[i];
    }
    if (t == 255) {
        offsets[nchunks] = 2 + part[255];
        meta[0] = 2 + part[255] + 2 + 4;
    }
}

// chunk c's bytes to stream[offsets[c] ..): whole words where the destination is aligned (two source words funnelled), bytes at the ends
__global__ void __launch_bounds__(256) png_gather_kernel(const uint8_t *__restrict__ scratch, const unsigned *__restrict__ sizes,
                                                          const unsigned long long *__restrict__ offsets, unsigned nchunks, uint8_t *__restrict__ stream)
{
    const unsigned c = blockIdx.x, t = threadIdx.x;
    const unsigned long long off = offsets[c];
    const unsigned nb = sizes[c];
    const uint8_t *src = scratch + (size_t)c * PNG_DEF_CAP;
    const unsigned head = min(nb, (unsigned)((4 - (off & 3)) & 3));
    if (t < head) stream[off + t] = src[t];
    const unsigned nwords = (nb - head) / 4;
    const unsigned sh = (head & 3) * 8;
    const unsigned *s32 = (const unsigned *)src;
    unsigned *d32 = (unsigned *)(stream + off + head);
    for (unsigned i = t; i < nwords; i += 256) {
        const unsigned lo = s32[i], hi = s32[i + 1];
        d32[i] = sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
    }
    const unsigned done = head + 4 * nwords;
    if (t < nb - done) stream[off + done + t] = src[done + t];
    if (c == 0 && t == 0) {
        stream[0] = 0x78;
        stream[1] = 0x01;
    }
    if (c == nchunks - 1 && t == 0) {
        uint8_t *e = stream + offsets[nchunks];
        e[0] = 0x03;          // BFINAL = 1, BTYPE = 01, the 7-bit end-of-block code 0000000
        e[1] = 0x00;
        e[2] = e[3] = e[4] = e[5] = 0;       // Adler-32: filled in by the host
    }
}

// ---- CRC-32 algebra on the host: the raw register after `len` more zero bytes, as a 32 x 32 matrix over GF(2) ----
static unsigned gf2_times(const unsigned *mat, unsigned vec)
{
    unsigned sum = 0;
    for (int i = 0; vec; vec >>= 1, ++i)
        if (vec & 1u) sum ^= mat[i];
    return sum;
}

static void gf2_square(unsigned *sq, const unsigned *mat)
{
    for (int n = 0; n < 32; ++n) sq[n] = gf2_times(mat, mat[n]);
}

// operator that advances a raw CRC register over `len` zero bytes
static void crc_zero_operator(unsigned long long len, unsigned *op)
{
    unsigned odd[32], even[32];
    odd[0] = 0xEDB88320u;                     // one zero BIT
    for (int n = 1; n < 32; ++n) odd[n] = 1u << (n - 1);
    gf2_square(even, odd);                    // 2 bits
    gf2_square(odd, even);                    // 4 bits
    for (int n = 0; n < 32; ++n) op[n] = 1u << n;       // identity
    unsigned cur[32], tmp[32];
    gf2_square(cur, odd);                     // 8 bits = one byte
    while (len) {
        if (len & 1ull) {
            for (int n = 0; n < 32; ++n) tmp[n] = gf2_times(cur, op[n]);
            memcpy(op, tmp, sizeof tmp);
        }
        len >>= 1;
        if (len) {
            gf2_square(tmp, cur);
            memcpy(cur, tmp, sizeof tmp);
        }
    }
}

static unsigned crc_table_byte(unsigned c, unsigned char b)
{
    c ^= b;
    for (int k = 0; k < 8; ++k) c = (c & 1u) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
    return c;
}

static void put_be32(unsigned char *p, unsigned v)
{
    p[0] = (unsigned char)(v >> 24);
    p[1] = (unsigned char)(v >> 16);
    p[2] = (unsigned char)(v >> 8);
    p[3] = (unsigned char)v;
}

static unsigned crc_bytes(const unsigned char *p, size_t n)
{
    unsigned c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; ++i) c = crc_table_byte(c, p[i]);
    return c ^ 0xFFFFFFFFu;
}


// ---- the literal/length code of one image (host) ----------------------------------------------------------------
// Huffman code lengths limited to `maxlen`: plain Huffman; while the deepest leaf is too deep the frequencies are halved
// (rounding up) and the tree rebuilt -- flattens the rare symbols, converges because equal frequencies give a balanced tree
static void huff_lengths(const unsigned long long *freq_in, int n, int maxlen, unsigned char *len_out)
{
    std::vector<unsigned long long> freq(freq_in, freq_in + n);
    for (;;) {
        std::vector<int> sym;
        for (int i = 0; i < n; ++i) {
            len_out[i] = 0;
            if (freq[i]) sym.push_back(i);
        }
        if (sym.empty()) return;
        if (sym.size() == 1) {
            len_out[sym[0]] = 1;
            return;
        }
        const int m = (int)sym.size();
        std::vector<unsigned long long> wt(2 * m);
        std::vector<int> parent(2 * m, -1);
        std::vector<int> order(sym);
        std::sort(order.begin(), order.end(), [&](int a, int b) { return freq[a] != freq[b] ? freq[a] < freq[b] : a < b; });
        for (int i = 0; i < m; ++i) wt[i] = freq[order[i]];
        // two queues: leaves (sorted) and internal nodes (created in non-decreasing weight)
        int leaf = 0, inode = m, made = m;
        auto take = [&]() {
            if (leaf < m && (inode >= made || wt[leaf] <= wt[inode])) return leaf++;
            return inode++;
        };
        while (made < 2 * m - 1) {
            const int a = take(), b = take();
            wt[made] = wt[a] + wt[b];
            parent[a] = parent[b] = made;
            ++made;
        }
        int deepest = 0;
        for (int i = 0; i < m; ++i) {
            int d = 0;
            for (int v = i; parent[v] >= 0; v = parent[v]) ++d;
            len_out[order[i]] = (unsigned char)(d > 255 ? 255 : d);
            deepest = d > deepest ? d : deepest;
        }
        if (deepest <= maxlen) return;
        for (int i = 0; i < n; ++i)
            if (freq[i]) freq[i] = (freq[i] + 1) >> 1;
    }
}

// canonical codes of deflate (RFC 1951 3.2.2), bit-reversed into stream order
static void huff_codes(const unsigned char *len, int n, unsigned *code_out)
{
    unsigned bl_count[16] = {0}, next_code[16] = {0};
    for (int i = 0; i < n; ++i) ++bl_count[len[i]];
    bl_count[0] = 0;
    unsigned code = 0;
    for (int b = 1; b < 16; ++b) {
        code = (code + bl_count[b - 1]) << 1;
        next_code[b] = code;
    }
    for (int i = 0; i < n; ++i) {
        code_out[i] = 0;
        if (!len[i]) continue;
        const unsigned c = next_code[len[i]]++;
        unsigned r = 0;
        for (int k = 0; k < len[i]; ++k)
            if (c & (1u << k)) r |= 1u << (len[i] - 1 - k);
        code_out[i] = r;
    }
}

struct bit_writer {
    unsigned w[PNG_HDR_WORDS] = {0};
    unsigned n = 0;
    bool overflow = false;
    void put(unsigned v, unsigned bits)
    {
        for (unsigned k = 0; k < bits; ++k, ++n) {
            if (n >= 32u * PNG_HDR_WORDS) {
                overflow = true;
                return;
            }
            if (v & (1u << k)) w[n >> 5] |= 1u << (n & 31);
        }
    }
};

static int png_make_codes(wfx_ctx *ctx, const unsigned *hist, unsigned nchunks, png_codes *pc)
{
    unsigned long long freq[286];
    for (int i = 0; i < 286; ++i) freq[i] = hist[i];
    freq[256] = nchunks;                       // one end-of-block per chunk
    unsigned char len[288] = {0};
    huff_lengths(freq, 286, 15, len);
    unsigned code[286];
    huff_codes(len, 286, code);
    memset(pc, 0, sizeof *pc);
    for (int v = 0; v < 256; ++v) pc->lit[v] = ((unsigned)len[v] << 24) | code[v];
    pc->eob = ((unsigned)len[256] << 24) | code[256];
    static const unsigned short base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const unsigned char extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    for (unsigned L = 3; L <= 258; ++L) {
        int sy = 28;
        while (base[sy] > L) --sy;
        if (L != 258 && sy == 28) sy = 27;
        const unsigned s = 257 + sy;
        if (!len[s]) {
            pc->mt[L - 3] = 0;                  // (no run of this length in the image: never looked up)
            continue;
        }
        const unsigned eb = extra[sy], ev = L - base[sy];
        // code, extra bits (LSB first), then the distance code: the single symbol 0, one bit, value 0
        const unsigned bits = code[s] | (ev << len[s]);
        const unsigned nb = len[s] + eb + 1;
        pc->mt[L - 3] = (nb << 24) | bits;
    }
    // header of every block
    int nlit = 286;
    while (nlit > 257 && !len[nlit - 1]) --nlit;
    unsigned char all[287];
    memcpy(all, len, nlit);
    all[nlit] = 1;                             // the distance code: one symbol of one bit
    unsigned long long clf[19] = {0};
    for (int i = 0; i <= nlit; ++i) ++clf[all[i]];
    unsigned char cll[19] = {0};
    huff_lengths(clf, 19, 7, cll);
    unsigned clc[19];
    huff_codes(cll, 19, clc);
    bit_writer bw;
    bw.put(0, 1);                              // BFINAL
    bw.put(2, 2);                              // BTYPE = 10: dynamic Huffman
    bw.put((unsigned)(nlit - 257), 5);
    bw.put(0, 5);                              // HDIST: 1 distance code
    bw.put(15, 4);                             // HCLEN: all 19 code length codes
    static const unsigned char ord[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    for (int i = 0; i < 19; ++i) bw.put(cll[ord[i]], 3);
    for (int i = 0; i <= nlit; ++i) bw.put(clc[all[i]], cll[all[i]]);      // (already in stream order)
    if (bw.overflow) return wfx_fail(ctx, WFX_ERR_STATE, "png: block header does not fit");
    pc->hdr_bits = bw.n;
    memcpy(pc->hdr, bw.w, sizeof bw.w);
    return 0;
}

// host part shared by both forms: Adler-32 from the row sums, CRC-32 from the span registers, the chunk headers around the stream
static void png_finish(unsigned char *h_stream, unsigned long long stream_len, const unsigned long long *h_sums, unsigned w, unsigned h, const unsigned *h_crc,
                       unsigned long long nspans)
{
    const unsigned row_bytes = w + 1;
    const unsigned long long raw = (unsigned long long)h * row_bytes;
    // Adler-32 of the h rows of n bytes each, in closed form: a = 1 + sum d_j, b = N + sum d_j (N - j) over stream positions j
    {
        const unsigned long long MOD = 65521ull, n = row_bytes, N = raw;
        unsigned long long a = 1, b = N % MOD;
        for (unsigned r = 0; r < h; ++r) {
            const unsigned long long S = h_sums[2 * r] % MOD, Wr = h_sums[2 * r + 1] % MOD;
            const unsigned long long after = (N - (unsigned long long)(r + 1) * n) % MOD;      // bytes behind this row
            a = (a + S) % MOD;
            b = (b + Wr + S * after) % MOD;
        }
        put_be32(h_stream + stream_len - 4, (unsigned)((b << 16) | a));
    }
    // CRC-32 of "IDAT" + stream: segments combined left to right, then the four Adler bytes
    unsigned crc;
    {
        static const unsigned char tag[4] = {'I', 'D', 'A', 'T'};
        unsigned reg = 0xFFFFFFFFu;
        for (int i = 0; i < 4; ++i) reg = crc_table_byte(reg, tag[i]);
        const unsigned long long body = stream_len - 4;
        unsigned op_full[32];
        crc_zero_operator(PNG_SEG, op_full);
        for (unsigned long long sp = 0; sp < nspans; ++sp) reg = gf2_times(op_full, reg) ^ h_crc[sp];
        static unsigned tab[256];
        static std::once_flag tab_once;
        std::call_once(tab_once, [] {
            for (unsigned i = 0; i < 256; ++i) tab[i] = crc_table_byte(0, (unsigned char)i);
        });
        for (unsigned long long i = nspans * PNG_SEG; i < body; ++i) reg = tab[(reg ^ h_stream[i]) & 255u] ^ (reg >> 8);
        for (int i = 0; i < 4; ++i) reg = crc_table_byte(reg, h_stream[body + i]);
        crc = reg ^ 0xFFFFFFFFu;
    }
    unsigned char *f = h_stream - 41;
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    memcpy(f, sig, 8);
    put_be32(f + 8, 13);
    memcpy(f + 12, "IHDR", 4);
    put_be32(f + 16, w);
    put_be32(f + 20, h);
    f[24] = 8;
    f[25] = 0;
    f[26] = 0;
    f[27] = 0;
    f[28] = 0;
    put_be32(f + 29, crc_bytes(f + 12, 17));
    put_be32(f + 33, (unsigned)stream_len);
    memcpy(f + 37, "IDAT", 4);
    unsigned char *t = h_stream + stream_len;
    put_be32(t, crc);
    put_be32(t + 4, 0);
    memcpy(t + 8, "IEND", 4);
    put_be32(t + 12, 0xAE426082u);
}

static int png_deflate(wfx_ctx *ctx, const void **file_bytes, size_t *nbytes);

extern "C" {

int wfx_decode_png(wfx_ctx *ctx, const void **file_bytes, size_t *nbytes)
{
    if (!ctx || !file_bytes || !nbytes) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null argument");
    (void)hipSetDevice(ctx->device);
    if (!ctx->ran) return wfx_fail(ctx, WFX_ERR_STATE, "no decode has run on this context");
    const auto te0 = std::chrono::steady_clock::now();
    WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const auto te1 = std::chrono::steady_clock::now();
    const unsigned w = (unsigned)ctx->dp.width, h = 4u * (unsigned)ctx->h_scal->height;
    if (ctx->h_scal->no_group || ctx->h_scal->nan_count || h == 0) return wfx_fail(ctx, WFX_ERR_STATE, "the decode produced no image");
    const uint8_t *img = ctx->img_in_ext ? (const uint8_t *)ctx->ext_img + 16 : (const uint8_t *)ctx->b_img.p;
    const unsigned row_bytes = w + 1;
    if (row_bytes > 65535) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "png: rows of %u bytes do not fit a stored deflate block", row_bytes);
    const unsigned rpb = 65535u / row_bytes;
    const unsigned nblocks = (h + rpb - 1) / rpb;
    const unsigned long long raw = (unsigned long long)h * row_bytes;
    const unsigned long long stream_len = 2ull + 5ull * nblocks + raw + 4ull;
    if (stream_len >= (1ull << 31)) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "png: image too large for one IDAT chunk");
    const unsigned long long nseg = (stream_len + PNG_SEG - 1) / PNG_SEG;
    // device: stream (16-byte aligned start), row sums, segment CRCs
    const size_t stream_cap = (size_t)((stream_len + 63) / 64 * 64);
    WFX_TRY(wfx_reserve(ctx, ctx->b_png, stream_cap + (size_t)h * 16 + (size_t)nseg * 4 + 256 + 1024 + 16));
    uint8_t *d_stream = (uint8_t *)ctx->b_png.p;
    unsigned long long *d_sums = (unsigned long long *)(d_stream + stream_cap);
    unsigned *d_crc = (unsigned *)(d_stream + stream_cap + (size_t)h * 16);
    WFX_LAUNCH(ctx, K_IMAGE, png_pack_kernel, dim3(wfx_stream_grid((stream_len + 3) / 4, 1024)), dim3(256), img, w, h, rpb, stream_len, (unsigned *)d_stream);
    WFX_LAUNCH(ctx, K_IMAGE, png_adler_rows_kernel, dim3((h + 3) / 4), dim3(256), img, w, h, d_sums);
    // whole spans of the body (everything but the four Adler bytes) on the device; the host finishes the tail (< 64 KiB)
    const unsigned long long body_len = stream_len - 4, nspans = body_len / PNG_SEG;
    static unsigned join_ops[8][32];
    static std::once_flag join_once;
    std::call_once(join_once, [] {
        for (int level = 0; level < 8; ++level) crc_zero_operator(256ull << level, join_ops[level]);
    });
    unsigned *d_ops = (unsigned *)(((uintptr_t)(d_crc + nseg) + 15) & ~(uintptr_t)15);
    WFX_HIP(ctx, hipMemcpyAsync(d_ops, join_ops, sizeof join_ops, hipMemcpyHostToDevice, ctx->stream));
    if (nspans) WFX_LAUNCH(ctx, K_IMAGE, png_crc_spans_kernel, dim3((unsigned)nspans), dim3(256), (const uint8_t *)d_stream, (const unsigned *)d_ops, d_crc);
    // pinned host image of the whole file: signature(8) IHDR(25) IDAT header(8) | stream | CRC(4) IEND(12).  The stream is placed
    // 16-byte aligned; the 41 bytes in front of it end right before it.
    const size_t lead = 48, file_len = 41 + (size_t)stream_len + 4 + 12;
    const size_t host_need = lead + stream_cap + 64 + (size_t)h * 16 + (size_t)nseg * 4;
    if (ctx->h_png_cap < host_need) {
        if (ctx->h_png) (void)hipHostFree(ctx->h_png);
        ctx->h_png = nullptr;
        ctx->h_png_cap = 0;
        if (hipHostMalloc((void **)&ctx->h_png, host_need, hipHostMallocDefault) != hipSuccess)
            return wfx_fail(ctx, WFX_ERR_OOM, "pinned host allocation of %zu bytes failed", host_need);
        ctx->h_png_cap = host_need;
    }
    unsigned char *hp = (unsigned char *)ctx->h_png;
    unsigned char *h_stream = hp + lead;
    unsigned long long *h_sums = (unsigned long long *)(hp + lead + stream_cap + 64);
    unsigned *h_crc = (unsigned *)((unsigned char *)h_sums + (size_t)h * 16);
    const bool dbg = getenv("WFX_DEBUG") != nullptr;
    const auto td0 = std::chrono::steady_clock::now();
    if (dbg) WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const auto td1 = std::chrono::steady_clock::now();
    WFX_HIP(ctx, hipMemcpyAsync(h_stream, d_stream, (size_t)stream_len, hipMemcpyDeviceToHost, ctx->stream));
    WFX_HIP(ctx, hipMemcpyAsync(h_sums, d_sums, (size_t)h * 16 + (size_t)nseg * 4, hipMemcpyDeviceToHost, ctx->stream));
    WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const auto td2 = std::chrono::steady_clock::now();
    png_finish(h_stream, stream_len, h_sums, w, h, h_crc, nspans);
    unsigned char *f = h_stream - 41;
    *file_bytes = f;
    *nbytes = file_len;
    if (dbg)
        fprintf(stderr, "[wfx] png: entry sync %.2f ms, launches %.2f ms, kernels %.2f ms, DMA of %.1f MB %.2f ms, check sums on the host %.2f ms\n",
                std::chrono::duration<double, std::milli>(te1 - te0).count(), std::chrono::duration<double, std::milli>(td0 - te1).count(),
                std::chrono::duration<double, std::milli>(td1 - td0).count(), stream_len / 1e6, std::chrono::duration<double, std::milli>(td2 - td1).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - td2).count());
    return 0;
}

int wfx_decode_png_ex(wfx_ctx *ctx, int deflate, const void **file_bytes, size_t *nbytes)
{
    return deflate ? png_deflate(ctx, file_bytes, nbytes) : wfx_decode_png(ctx, file_bytes, nbytes);
}

int wfx_decode_save_png(wfx_ctx *ctx, const char *path, size_t *bytes_written) { return wfx_decode_save_png_ex(ctx, path, 0, bytes_written); }

int wfx_decode_save_png_ex(wfx_ctx *ctx, const char *path, int deflate, size_t *bytes_written)
{
    if (!path) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null path");
    const void *p = nullptr;
    size_t n = 0;
    const auto tp0 = std::chrono::steady_clock::now();
    WFX_TRY(wfx_decode_png_ex(ctx, deflate, &p, &n));
    const auto tp1 = std::chrono::steady_clock::now();
    // the file image sits in pinned memory; copying 27 MB into the page cache is memcpy-bound per thread (8 ms for one
    // writer), so the file is written in slices by a few threads
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "cannot open %s for writing", path);
    static const size_t slice = getenv("WFX_PNG_SLICE_KB") ? (size_t)atol(getenv("WFX_PNG_SLICE_KB")) << 10 : (size_t)4 << 20;
    unsigned nthr = std::thread::hardware_concurrency(), cap = getenv("WFX_PNG_THREADS") ? (unsigned)atoi(getenv("WFX_PNG_THREADS")) : 8u;
    nthr = nthr < 1 ? 1 : nthr > cap ? cap : nthr;
    const size_t nslices = (n + slice - 1) / slice;
    if (nthr > nslices) nthr = (unsigned)nslices;
    std::atomic<size_t> next{0};
    std::atomic<bool> failed{false};
    // (measured on the 79 MB file of the 60-minute 48 kHz picture: 15-27 ms whether 8 or 19 threads write, and the same through a shared
    // mapping of the pre-sized file -- the time is the kernel's page allocation for the new file, not the copy)
    auto writer = [&]() {
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= nslices || failed.load()) return;
            size_t off = k * slice;
            const size_t end = off + slice < n ? off + slice : n;
            while (off < end) {
                const ssize_t wr = pwrite(fd, (const unsigned char *)p + off, end - off, (off_t)off);
                if (wr <= 0) {
                    failed.store(true);
                    return;
                }
                off += (size_t)wr;
            }
        }
    };
    std::vector<std::thread> pool;
    for (unsigned i = 1; i < nthr; ++i) pool.emplace_back(writer);
    writer();
    for (auto &th : pool) th.join();
    const int rc = close(fd);
    if (failed.load() || rc != 0) return wfx_fail(ctx, WFX_ERR_STATE, "short write to %s", path);
    if (getenv("WFX_DEBUG"))
        fprintf(stderr, "[wfx] png: assemble + DMA %.2f ms, write (%u threads) %.2f ms\n", std::chrono::duration<double, std::milli>(tp1 - tp0).count(),
                nthr, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp1).count());
    if (bytes_written) *bytes_written = n;
    return 0;
}

}  // extern "C"

static int png_deflate(wfx_ctx *ctx, const void **file_bytes, size_t *nbytes)
{
    if (!ctx || !file_bytes || !nbytes) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null argument");
    (void)hipSetDevice(ctx->device);
    if (!ctx->ran) return wfx_fail(ctx, WFX_ERR_STATE, "no decode has run on this context");
    const auto t0 = std::chrono::steady_clock::now();
    WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const unsigned w = (unsigned)ctx->dp.width, h = 4u * (unsigned)ctx->h_scal->height;
    if (ctx->h_scal->no_group || ctx->h_scal->nan_count || h == 0) return wfx_fail(ctx, WFX_ERR_STATE, "the decode produced no image");
    const uint8_t *img = ctx->img_in_ext ? (const uint8_t *)ctx->ext_img + 16 : (const uint8_t *)ctx->b_img.p;
    const unsigned row_bytes = w + 1;
    if (row_bytes > PNG_DEF_MAXRAW) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "png: rows of %u bytes do not fit a chunk of the device encoder", row_bytes);
    unsigned rpc = PNG_DEF_MAXRAW / row_bytes;
    if (rpc > 64) rpc = 64;
    const unsigned nchunks = (h + rpc - 1) / rpc;
    const unsigned long long raw = (unsigned long long)h * row_bytes;
    const unsigned long long max_len = 2ull + raw + 5ull * nchunks + 2ull + 4ull;           // every chunk stored
    if (max_len >= (1ull << 31)) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "png: image too large for one IDAT chunk");
    const unsigned long long max_seg = (max_len + PNG_SEG - 1) / PNG_SEG;
    // device: stream | row sums | span CRCs | join operators | chunk scratch | sizes | offsets | histogram | codes | meta
    const size_t stream_cap = (size_t)((max_len + 63) / 64 * 64);
    size_t o = stream_cap;
    const size_t o_sums = o;      o += (size_t)h * 16;
    const size_t o_crc = o;       o += ((size_t)max_seg * 4 + 15) / 16 * 16;
    const size_t o_ops = o;       o += 1024;
    const size_t o_scratch = o;   o += (size_t)nchunks * PNG_DEF_CAP + 64;
    const size_t o_sizes = o;     o += ((size_t)nchunks * 4 + 15) / 16 * 16;
    const size_t o_offs = o;      o += ((size_t)(nchunks + 1) * 8 + 15) / 16 * 16;
    const size_t o_hist = o;      o += 288 * 4;
    const size_t o_codes = o;     o += (sizeof(png_codes) + 15) / 16 * 16;
    const size_t o_meta = o;      o += 16;
    WFX_TRY(wfx_reserve(ctx, ctx->b_png, o + 64));
    uint8_t *base = (uint8_t *)ctx->b_png.p;
    uint8_t *d_stream = base;
    unsigned long long *d_sums = (unsigned long long *)(base + o_sums);
    unsigned *d_crc = (unsigned *)(base + o_crc);
    unsigned *d_ops = (unsigned *)(base + o_ops);
    uint8_t *d_scratch = base + o_scratch;
    unsigned *d_sizes = (unsigned *)(base + o_sizes);
    unsigned long long *d_offs = (unsigned long long *)(base + o_offs);
    unsigned *d_hist = (unsigned *)(base + o_hist);
    png_codes *d_codes = (png_codes *)(base + o_codes);
    unsigned long long *d_meta = (unsigned long long *)(base + o_meta);
    // pinned host: the file image as in the stored form, then the row sums, the span CRCs and a small staging area
    const size_t lead = 48;
    const size_t h_sums_off = lead + stream_cap + 64, h_crc_off = h_sums_off + (size_t)h * 16, h_stage_off = h_crc_off + ((size_t)max_seg * 4 + 15) / 16 * 16;
    const size_t host_need = h_stage_off + 288 * 4 + sizeof(png_codes) + 64;
    if (ctx->h_png_cap < host_need) {
        if (ctx->h_png) (void)hipHostFree(ctx->h_png);
        ctx->h_png = nullptr;
        ctx->h_png_cap = 0;
        if (hipHostMalloc((void **)&ctx->h_png, host_need, hipHostMallocDefault) != hipSuccess)
            return wfx_fail(ctx, WFX_ERR_OOM, "pinned host allocation of %zu bytes failed", host_need);
        ctx->h_png_cap = host_need;
    }
    unsigned char *hp = (unsigned char *)ctx->h_png;
    unsigned char *h_stream = hp + lead;
    unsigned long long *h_sums = (unsigned long long *)(hp + h_sums_off);
    unsigned *h_crc = (unsigned *)(hp + h_crc_off);
    unsigned *h_hist = (unsigned *)(hp + h_stage_off);
    png_codes *h_codes = (png_codes *)(hp + h_stage_off + 288 * 4);
    unsigned long long *h_meta = (unsigned long long *)((unsigned char *)h_codes + sizeof(png_codes));
    // 1. token histogram (+ Adler row sums) -> host -> the image's code -> device
    WFX_HIP(ctx, hipMemsetAsync(d_hist, 0, 288 * 4, ctx->stream));
    WFX_LAUNCH(ctx, K_IMAGE, png_hist_kernel, dim3(nchunks), dim3(256), img, w, h, rpc, d_hist, d_sums);
    WFX_HIP(ctx, hipMemcpyAsync(h_hist, d_hist, 288 * 4, hipMemcpyDeviceToHost, ctx->stream));
    WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const auto t1 = std::chrono::steady_clock::now();
    WFX_TRY(png_make_codes(ctx, h_hist, nchunks, h_codes));
    WFX_HIP(ctx, hipMemcpyAsync(d_codes, h_codes, sizeof(png_codes), hipMemcpyHostToDevice, ctx->stream));
    // 2. chunks -> scratch, sizes -> offsets
    WFX_LAUNCH(ctx, K_IMAGE, png_encode_kernel, dim3(nchunks), dim3(256), img, w, h, rpc, (const png_codes *)d_codes, d_scratch, d_sizes);
    WFX_LAUNCH(ctx, K_IMAGE, png_scan_kernel, dim3(1), dim3(256), (const unsigned *)d_sizes, nchunks, d_offs, d_meta);
    WFX_LAUNCH(ctx, K_IMAGE, png_gather_kernel, dim3(nchunks), dim3(256), (const uint8_t *)d_scratch, (const unsigned *)d_sizes, (const unsigned long long *)d_offs,
               nchunks, d_stream);
    WFX_HIP(ctx, hipMemcpyAsync(h_meta, d_meta, 8, hipMemcpyDeviceToHost, ctx->stream));
    WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const auto t2 = std::chrono::steady_clock::now();
    const unsigned long long stream_len = h_meta[0];
    if (stream_len < 8 || stream_len > max_len) return wfx_fail(ctx, WFX_ERR_STATE, "png: encoder produced %llu bytes (at most %llu expected)", stream_len, max_len);
    // 3. CRC spans of the finished stream, then everything to the host
    const unsigned long long body_len = stream_len - 4, nspans = body_len / PNG_SEG;
    static unsigned join_ops[8][32];
    static std::once_flag join_once;
    std::call_once(join_once, [] {
        for (int level = 0; level < 8; ++level) crc_zero_operator(256ull << level, join_ops[level]);
    });
    WFX_HIP(ctx, hipMemcpyAsync(d_ops, join_ops, sizeof join_ops, hipMemcpyHostToDevice, ctx->stream));
    if (nspans) WFX_LAUNCH(ctx, K_IMAGE, png_crc_spans_kernel, dim3((unsigned)nspans), dim3(256), (const uint8_t *)d_stream, (const unsigned *)d_ops, d_crc);
    WFX_HIP(ctx, hipMemcpyAsync(h_stream, d_stream, (size_t)stream_len, hipMemcpyDeviceToHost, ctx->stream));
    WFX_HIP(ctx, hipMemcpyAsync(h_sums, d_sums, (size_t)h * 16, hipMemcpyDeviceToHost, ctx->stream));
    if (nspans) WFX_HIP(ctx, hipMemcpyAsync(h_crc, d_crc, (size_t)nspans * 4, hipMemcpyDeviceToHost, ctx->stream));
    WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const auto t3 = std::chrono::steady_clock::now();
    png_finish(h_stream, stream_len, h_sums, w, h, h_crc, nspans);
    *file_bytes = h_stream - 41;
    *nbytes = 41 + (size_t)stream_len + 4 + 12;
    if (getenv("WFX_DEBUG"))
        fprintf(stderr, "[wfx] png (deflate): histogram %.2f ms, code + encode + gather %.2f ms, CRC + DMA of %.1f MB (%.1f MB filtered) %.2f ms, check sums on the host %.2f ms\n",
                std::chrono::duration<double, std::milli>(t1 - t0).count(), std::chrono::duration<double, std::milli>(t2 - t1).count(), stream_len / 1e6, raw / 1e6,
                std::chrono::duration<double, std::milli>(t3 - t2).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t3).count());
    return 0;
}
