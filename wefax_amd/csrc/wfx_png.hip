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
    *file_bytes = f;
    *nbytes = file_len;
    if (dbg)
        fprintf(stderr, "[wfx] png: entry sync %.2f ms, launches %.2f ms, kernels %.2f ms, DMA of %.1f MB %.2f ms, check sums on the host %.2f ms\n",
                std::chrono::duration<double, std::milli>(te1 - te0).count(), std::chrono::duration<double, std::milli>(td0 - te1).count(),
                std::chrono::duration<double, std::milli>(td1 - td0).count(), stream_len / 1e6, std::chrono::duration<double, std::milli>(td2 - td1).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - td2).count());
    return 0;
}

int wfx_decode_save_png(wfx_ctx *ctx, const char *path, size_t *bytes_written)
{
    if (!path) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null path");
    const void *p = nullptr;
    size_t n = 0;
    const auto tp0 = std::chrono::steady_clock::now();
    WFX_TRY(wfx_decode_png(ctx, &p, &n));
    const auto tp1 = std::chrono::steady_clock::now();
    // the file image sits in pinned memory; copying 27 MB into the page cache is memcpy-bound per thread (8 ms for one
    // writer), so the file is written in slices by a few threads
    const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "cannot open %s for writing", path);
    const size_t slice = (size_t)4 << 20;
    unsigned nthr = std::thread::hardware_concurrency();
    nthr = nthr < 1 ? 1 : nthr > 8 ? 8 : nthr;
    const size_t nslices = (n + slice - 1) / slice;
    if (nthr > nslices) nthr = (unsigned)nslices;
    std::atomic<size_t> next{0};
    std::atomic<bool> failed{false};
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
