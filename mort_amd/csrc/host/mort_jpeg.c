/*
 * mort_jpeg.c -- baseline (SOF0, 8-bit, Huffman, sequential) JPEG decoder for the image_texture input of scenes 3, 8
 * and 9: the reference loads imgs/earthmap.jpg through its vendored stb_image (img_loader.h:38-44, textures.cuh:89-127)
 * and uploads the decoded RGB bytes.  JPEG decoding is integer arithmetic, so the texels can be reproduced exactly:
 * this decoder follows the same published pipeline -- ITU T.81 Huffman decoding (Annex F), dequantisation, the
 * libjpeg "jidctint / DCT_ISLOW" integer inverse DCT with 12-bit constants and the rounding points stb_image uses
 * (stb_image.h:2424-2508: column pass keeps 2 extra bits, row pass removes 17 and adds the 128 level shift), and the
 * 20-bit fixed-point YCbCr -> RGB conversion (stb_image.h:3657-3672) -- written from scratch here.  Pinned byte for byte
 * against tests/golden/earthmap_rgb.npz (the reference's own stb_image output) by tests/test_jpeg.py.
 *
 * Supported: 1 or 3 components, any sampling factors with nearest (box) replication -- the only smoothing-free choice
 * that needs no parity claim: earthmap.jpg is 1x1 sampled, for which every decoder's upsampling is the identity.
 * Subsampled files decode, but their chroma upsampling is not stb_image's (it uses a triangle filter) and says so.
 * Restart intervals are honoured.  Progressive / arithmetic / 12-bit files are rejected.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mort_host.h"

typedef struct {
    /* T.81 F.2.2.3: codes of each length are consecutive; a code of length l is valid iff code <= maxcode[l] */
    int mincode[17], maxcode[18], valptr[17];
    uint8_t vals[256];
    int present;
} huff_table;

typedef struct {
    int id, h, v, tq, td, ta;
    int dc_pred;
    int bw, bh;        /* blocks per row / column, padded to whole MCUs */
    uint8_t *plane;    /* bw*8 x bh*8 samples */
} component;

typedef struct {
    const uint8_t *p, *end;
    uint32_t bitbuf;
    int bitcnt;
    int marker;        /* marker met inside entropy-coded data, or 0 */
} bitreader;

static int build_table(huff_table *t, const uint8_t counts[16], const uint8_t *vals, int nvals) {
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
        t->valptr[l] = k;
        t->mincode[l] = code;
        code += counts[l - 1];
        k += counts[l - 1];
        t->maxcode[l] = counts[l - 1] ? code - 1 : -1;
        if (code > (1 << l)) return -1;
        code <<= 1;
    }
    t->maxcode[17] = 0x7fffffff;
    if (k != nvals || k > 256) return -1;
    memcpy(t->vals, vals, (size_t)nvals);
    t->present = 1;
    return 0;
}

static int next_bit(bitreader *b) {
    if (b->bitcnt == 0) {
        int byte = 0;
        if (b->marker == 0 && b->p < b->end) {
            byte = *b->p++;
            if (byte == 0xff) {
                int c = (b->p < b->end) ? *b->p : 0xd9;
                if (c == 0) b->p++;              /* stuffed zero */
                else { b->marker = c; byte = 0; b->p++; } /* a marker ends the segment: feed zeros */
            }
        }
        b->bitbuf = (uint32_t)byte;
        b->bitcnt = 8;
    }
    b->bitcnt--;
    return (int)((b->bitbuf >> b->bitcnt) & 1u);
}
static int receive(bitreader *b, int n) { int v = 0; while (n-- > 0) v = (v << 1) | next_bit(b); return v; }
/* T.81 F.2.2.1 EXTEND */
static int extend(int v, int t) { return (t > 0 && v < (1 << (t - 1))) ? v - (1 << t) + 1 : v; }
static int decode_symbol(bitreader *b, const huff_table *t) {
    int code = 0;
    for (int l = 1; l <= 16; l++) {
        code = (code << 1) | next_bit(b);
        if (t->maxcode[l] >= 0 && code <= t->maxcode[l] && code >= t->mincode[l]) return t->vals[t->valptr[l] + code - t->mincode[l]];
    }
    return -1;
}

/* 64 entries + 15: a damaged scan can run a block's coefficient index up to 63 + 15; the reference's decoder lets those land on
 * coefficient 63 instead of refusing the file (external/stb_image.h:2253-2256 with its padded table), and so does this one */
static const uint8_t zigzag[64 + 15] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                        35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
                                        63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};
/* the two range checks the reference's decoder applies to a block's DC term (external/stb_image.h:2221-2224, 1068-1082): the prediction
 * must not overflow int, and the dequantised value -- both factors first narrowed to 16 bits -- must stay a short, judged the way that
 * decoder judges it (it accepts a factor of 0 or -1 outright and compares against the truncated quotient otherwise) */
static int dc_sum_ok(int pred, int diff) {
    const long long v = (long long)pred + (long long)diff;
    return v >= -2147483647LL - 1 && v <= 2147483647LL;
}
static int dc_product_ok(short dc, short q) {
    if (q == 0 || q == -1) return 1;
    if ((dc >= 0) == (q >= 0)) return dc <= 32767 / q;
    if (q < 0) return dc <= -32768 / q;
    return dc >= -32768 / q;
}

/* 12-bit fixed-point constant of a float literal, rounded as (int)(x * 4096 + 0.5) */
#define FX(x) ((int)(((x) * 4096 + 0.5)))

/* one 8-point inverse DCT (jidctint, "islow"): even part into e[0..3], odd part into o[0..3], all scaled by 4096 */
static void idct8(const int s[8], int e[4], int o[4]) {
    int p1 = (s[2] + s[6]) * FX(0.5411961f);
    const int t2 = p1 + s[6] * FX(-1.847759065f);
    const int t3 = p1 + s[2] * FX(0.765366865f);
    const int t0 = (s[0] + s[4]) * 4096, t1 = (s[0] - s[4]) * 4096;
    e[0] = t0 + t3; e[3] = t0 - t3; e[1] = t1 + t2; e[2] = t1 - t2;
    int a0 = s[7], a1 = s[5], a2 = s[3], a3 = s[1];
    int p3 = a0 + a2, p4 = a1 + a3;
    int q1 = a0 + a3, q2 = a1 + a2;
    const int p5 = (p3 + p4) * FX(1.175875602f);
    a0 = a0 * FX(0.298631336f);
    a1 = a1 * FX(2.053119869f);
    a2 = a2 * FX(3.072711026f);
    a3 = a3 * FX(1.501321110f);
    q1 = p5 + q1 * FX(-0.899976223f);
    q2 = p5 + q2 * FX(-2.562915447f);
    p3 = p3 * FX(-1.961570560f);
    p4 = p4 * FX(-0.390180644f);
    o[3] = a3 + q1 + p4; /* pairs with e[0] */
    o[2] = a2 + q2 + p3; /* pairs with e[1] */
    o[1] = a1 + q2 + p4; /* pairs with e[2] */
    o[0] = a0 + q1 + p3; /* pairs with e[3] */
}
static uint8_t clamp255(int x) { return (uint8_t)(x < 0 ? 0 : x > 255 ? 255 : x); }

/* dequantised coefficients (natural order, already narrowed to 16 bits as the reference's decoder stores them) -> 8x8 samples */
static void idct_block(const short *d, uint8_t *out, int stride) {
    int mid[64];
    for (int c = 0; c < 8; c++) { /* columns: keep 2 extra bits (>> 10 after the 4096 scale, rounding constant 512) */
        int s[8], e[4], o[4];
        for (int r = 0; r < 8; r++) s[r] = d[r * 8 + c];
        idct8(s, e, o);
        for (int k = 0; k < 4; k++) {
            const int x = e[k] + 512;
            mid[k * 8 + c] = (x + o[3 - k]) >> 10;
            mid[(7 - k) * 8 + c] = (x - o[3 - k]) >> 10;
        }
    }
    for (int r = 0; r < 8; r++) { /* rows: remove 2^17, round, add the level shift of 128 */
        int e[4], o[4];
        idct8(mid + r * 8, e, o);
        uint8_t *px = out + (size_t)r * stride;
        for (int k = 0; k < 4; k++) {
            const int x = e[k] + 65536 + (128 << 17);
            px[k] = clamp255((x + o[3 - k]) >> 17);
            px[7 - k] = clamp255((x - o[3 - k]) >> 17);
        }
    }
}

#define CFIX(x) (((int)((x) * 4096.0f + 0.5f)) << 8)
static void ycc_to_rgb(uint8_t *out, int y, int cb, int cr) {
    const int yf = (y << 20) + (1 << 19);
    cb -= 128; cr -= 128;
    int r = yf + cr * CFIX(1.40200f);
    int g = yf + (cr * -CFIX(0.71414f)) + ((cb * -CFIX(0.34414f)) & (int)0xffff0000);
    int b = yf + cb * CFIX(1.77200f);
    out[0] = clamp255(r >> 20); out[1] = clamp255(g >> 20); out[2] = clamp255(b >> 20);
}

static unsigned rd16(const uint8_t *p) { return ((unsigned)p[0] << 8) | p[1]; }

unsigned char *mort_decode_jpeg(const unsigned char *data, size_t size, int *width, int *height) {
    if (!data || size < 4 || data[0] != 0xff || data[1] != 0xd8) return NULL;
    uint16_t quant[4][64];
    int have_q[4] = {0, 0, 0, 0};
    huff_table hdc[4], hac[4];
    memset(hdc, 0, sizeof hdc); memset(hac, 0, sizeof hac);
    component comp[3];
    memset(comp, 0, sizeof comp);
    int ncomp = 0, W = 0, H = 0, hmax = 1, vmax = 1, restart = 0, have_frame = 0;
    uint8_t *rgb = NULL;
    const uint8_t *p = data + 2, *end = data + size;
    int ok = 0;

    while (p + 4 <= end) {
        if (p[0] != 0xff) { p++; continue; }
        const int m = p[1];
        if (m == 0xff) { p++; continue; }
        p += 2;
        if (m == 0xd9) break;
        if (m == 0x01 || (m >= 0xd0 && m <= 0xd7)) continue;
        if (p + 2 > end) goto fail;
        const unsigned len = rd16(p);
        if (len < 2 || p + len > end) goto fail;
        const uint8_t *seg = p + 2, *seg_end = p + len;
        if (m == 0xdb) { /* DQT */
            while (seg < seg_end) {
                const int pq = seg[0] >> 4, tq = seg[0] & 15;
                if (tq > 3 || seg + 1 + (pq ? 128 : 64) > seg_end) goto fail;
                for (int i = 0; i < 64; i++) quant[tq][zigzag[i]] = pq ? (uint16_t)rd16(seg + 1 + 2 * i) : seg[1 + i];
                have_q[tq] = 1;
                seg += 1 + (pq ? 128 : 64);
            }
        } else if (m == 0xc4) { /* DHT */
            while (seg + 17 <= seg_end) {
                const int tc = seg[0] >> 4, th = seg[0] & 15;
                int n = 0;
                for (int i = 0; i < 16; i++) n += seg[1 + i];
                if (tc > 1 || th > 3 || seg + 17 + n > seg_end) goto fail;
                if (build_table(tc ? &hac[th] : &hdc[th], seg + 1, seg + 17, n) != 0) goto fail;
                seg += 17 + n;
            }
        } else if (m == 0xc0 || m == 0xc1) { /* SOF0 / SOF1 with 8-bit samples: sequential Huffman */
            if (seg + 6 > seg_end || seg[0] != 8) goto fail;
            H = (int)rd16(seg + 1); W = (int)rd16(seg + 3); ncomp = seg[5];
            if ((ncomp != 1 && ncomp != 3) || W <= 0 || H <= 0 || seg + 6 + 3 * ncomp > seg_end) goto fail;
            for (int i = 0; i < ncomp; i++) {
                comp[i].id = seg[6 + 3 * i]; comp[i].h = seg[7 + 3 * i] >> 4; comp[i].v = seg[7 + 3 * i] & 15; comp[i].tq = seg[8 + 3 * i];
                if (comp[i].h < 1 || comp[i].h > 4 || comp[i].v < 1 || comp[i].v > 4 || comp[i].tq > 3) goto fail;
                if (comp[i].h > hmax) hmax = comp[i].h;
                if (comp[i].v > vmax) vmax = comp[i].v;
            }
            have_frame = 1;
        } else if (m == 0xc2 || (m >= 0xc3 && m <= 0xcf && m != 0xc4 && m != 0xc8 && m != 0xcc)) {
            goto fail; /* progressive, lossless, arithmetic: not this texture path */
        } else if (m == 0xdd) { /* DRI */
            if (seg + 2 > seg_end) goto fail;
            restart = (int)rd16(seg);
        } else if (m == 0xda) { /* SOS: the one scan of a baseline file */
            if (!have_frame || seg + 1 > seg_end || seg[0] != ncomp || seg + 1 + 2 * ncomp + 3 > seg_end) goto fail;
            for (int i = 0; i < ncomp; i++) {
                int k = -1;
                for (int j = 0; j < ncomp; j++) if (comp[j].id == seg[1 + 2 * i]) k = j;
                if (k != i) goto fail; /* components in frame order */
                comp[k].td = seg[2 + 2 * i] >> 4; comp[k].ta = seg[2 + 2 * i] & 15;
                if (comp[k].td > 3 || comp[k].ta > 3 || !hdc[comp[k].td].present || !hac[comp[k].ta].present || !have_q[comp[k].tq]) goto fail;
            }
            const int mcu_w = 8 * hmax, mcu_h = 8 * vmax;
            const int mcus_x = (W + mcu_w - 1) / mcu_w, mcus_y = (H + mcu_h - 1) / mcu_h;
            for (int i = 0; i < ncomp; i++) {
                comp[i].bw = mcus_x * comp[i].h; comp[i].bh = mcus_y * comp[i].v;
                comp[i].plane = (uint8_t *)malloc((size_t)comp[i].bw * 8 * (size_t)comp[i].bh * 8);
                if (!comp[i].plane) goto fail;
                comp[i].dc_pred = 0;
            }
            bitreader br;
            br.p = seg_end; br.end = end; br.bitbuf = 0; br.bitcnt = 0; br.marker = 0;
            int until_restart = restart;
            for (int my = 0; my < mcus_y; my++)
                for (int mx = 0; mx < mcus_x; mx++) {
                    if (restart && until_restart == 0) { /* RSTn: byte-align, reset predictions */
                        br.bitcnt = 0;
                        if (br.marker >= 0xd0 && br.marker <= 0xd7) br.marker = 0;
                        else if (br.p + 2 <= br.end && br.p[0] == 0xff && br.p[1] >= 0xd0 && br.p[1] <= 0xd7) br.p += 2;
                        for (int i = 0; i < ncomp; i++) comp[i].dc_pred = 0;
                        until_restart = restart;
                    }
                    for (int i = 0; i < ncomp; i++)
                        for (int by = 0; by < comp[i].v; by++)
                            for (int bx = 0; bx < comp[i].h; bx++) {
                                short blk[64];
                                memset(blk, 0, sizeof blk);
                                const uint16_t *q = quant[comp[i].tq];
                                int t = decode_symbol(&br, &hdc[comp[i].td]);
                                if (t < 0 || t > 15) goto fail;
                                const int diff = t ? extend(receive(&br, t), t) : 0;
                                if (!dc_sum_ok(comp[i].dc_pred, diff)) goto fail;
                                comp[i].dc_pred += diff;
                                if (!dc_product_ok((short)comp[i].dc_pred, (short)q[0])) goto fail;
                                blk[0] = (short)(comp[i].dc_pred * q[0]);
                                for (int k = 1; k < 64;) {
                                    const int rs = decode_symbol(&br, &hac[comp[i].ta]);
                                    if (rs < 0) goto fail;
                                    const int r = rs >> 4, s = rs & 15;
                                    if (s == 0) { if (r != 15) break; k += 16; continue; }
                                    k += r; /* <= 63 + 15: the padded table */
                                    const int z = zigzag[k++];
                                    blk[z] = (short)(extend(receive(&br, s), s) * q[z]);
                                }
                                const int X = (mx * comp[i].h + bx) * 8, Y = (my * comp[i].v + by) * 8;
                                idct_block(blk, comp[i].plane + (size_t)Y * comp[i].bw * 8 + X, comp[i].bw * 8);
                            }
                    if (restart) until_restart--;
                }
            /* planes -> RGB (nearest replication of subsampled components: identity for 1x1 files such as earthmap.jpg) */
            rgb = (uint8_t *)malloc((size_t)W * H * 3);
            if (!rgb) goto fail;
            for (int y = 0; y < H; y++)
                for (int x = 0; x < W; x++) {
                    int s[3] = {0, 128, 128};
                    for (int i = 0; i < ncomp; i++) {
                        const int sx = x * comp[i].h / hmax, sy = y * comp[i].v / vmax;
                        s[i] = comp[i].plane[(size_t)sy * comp[i].bw * 8 + sx];
                    }
                    uint8_t *o = rgb + ((size_t)y * W + x) * 3;
                    if (ncomp == 1) o[0] = o[1] = o[2] = (uint8_t)s[0];
                    else ycc_to_rgb(o, s[0], s[1], s[2]);
                }
            ok = 1;
            break;
        }
        p += len;
    }
fail:
    for (int i = 0; i < 3; i++) free(comp[i].plane);
    if (!ok) { free(rgb); return NULL; }
    *width = W; *height = H;
    return rgb;
}

unsigned char *mort_read_jpeg(const char *path, int *width, int *height) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    unsigned char *buf = NULL, *out = NULL;
    if (fseek(f, 0, SEEK_END) == 0) {
        const long n = ftell(f);
        if (n > 0 && n < (1l << 30) && fseek(f, 0, SEEK_SET) == 0 && (buf = (unsigned char *)malloc((size_t)n)) != NULL && fread(buf, 1, (size_t)n, f) == (size_t)n)
            out = mort_decode_jpeg(buf, (size_t)n, width, height);
    }
    free(buf);
    fclose(f);
    return out;
}

/* .jpg / .jpeg -> mort_read_jpeg, anything else -> mort_read_ppm (decoded-texel fixtures) */
unsigned char *mort_read_image(const char *path, int *width, int *height) {
    const char *dot = path ? strrchr(path, '.') : NULL;
    if (dot && (strcmp(dot, ".jpg") == 0 || strcmp(dot, ".jpeg") == 0 || strcmp(dot, ".JPG") == 0)) return mort_read_jpeg(path, width, height);
    return mort_read_ppm(path, width, height);
}
