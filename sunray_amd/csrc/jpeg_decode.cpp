// JPEG (ITU-T T.81 / ISO 10918-1: baseline, extended sequential and progressive DCT, Huffman-coded, 8-bit) -> 8-bit pixels for the
// glTF loader.
//
// The reference gets its textures from `gltf::import` (gltf/mod.rs:57-67), which decodes JPEG through the `image` crate
// (0.25: zune-jpeg) into R8 (greyscale) or R8G8B8 — formats Image::new_from_data accepts (image/mod.rs:98-104). The
// decoder library is a Cargo dependency that is not under /root/reference; JPEG leaves the inverse DCT's rounding and the
// chroma upsampling filter to the implementation, so the decoded bytes of two conforming decoders differ by a few
// units: PARITY UNPINNED for JPEG texels. This one is the standard's own definition: dequantise, the separable
// 8x8 inverse DCT of Annex A.3.3 evaluated in double precision, level shift + round to nearest + clamp, the IJG library's
// triangle filters for 2:1 subsampled chroma (replication for other ratios), JFIF YCbCr -> RGB. tests/test_gltf.py checks it against libjpeg (Pillow) within that margin.
// Arithmetic-coded, lossless, hierarchical, 12-bit and 4-component (CMYK) files are refused.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "host.h"

namespace srh {
namespace {

const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                             30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huffman {            // Annex C / F.2.2.3: codes of each length are consecutive; decode bit by bit against maxcode
    bool present = false;
    uint8_t vals[256];
    int mincode[17], maxcode[18], valptr[17];
    void build(const uint8_t counts[16], const uint8_t* symbols) {
        int code = 0, k = 0;
        for (int len = 1; len <= 16; len++) {
            valptr[len] = k;
            mincode[len] = code;
            code += counts[len - 1];
            k += counts[len - 1];
            maxcode[len] = counts[len - 1] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
        memcpy(vals, symbols, (size_t)k);
        present = true;
    }
};

struct BitReader {          // entropy-coded segment: 0xFF00 is a stuffed 0xFF, any other marker ends the data (zeros are fed)
    const uint8_t* d; size_t n, pos; uint32_t acc = 0; int cnt = 0; bool hit_marker = false, ran_out = false;
    int bit() {
        if (cnt == 0) {
            uint8_t b = 0;
            if (!hit_marker && pos < n) {
                b = d[pos];
                if (b == 0xFF) {
                    if (pos + 1 < n && d[pos + 1] == 0x00) pos += 2;
                    else { hit_marker = true; b = 0; }
                } else pos++;
            } else { hit_marker = true; ran_out = ran_out || pos >= n; }
            acc = b; cnt = 8;
        }
        cnt--;
        return (int)((acc >> cnt) & 1u);
    }
    int bits(int k) { int v = 0; while (k-- > 0) v = (v << 1) | bit(); return v; }
    void align() { cnt = 0; }
};

int decode_symbol(BitReader& br, const Huffman& h, bool& ok) {
    int code = 0;
    for (int len = 1; len <= 16; len++) {
        code = (code << 1) | br.bit();
        if (h.maxcode[len] >= 0 && code <= h.maxcode[len] && code >= h.mincode[len]) return h.vals[h.valptr[len] + code - h.mincode[len]];
    }
    ok = false;
    return 0;
}
int extend(int v, int s) { return (s && v < (1 << (s - 1))) ? v - (1 << s) + 1 : v; }   // F.2.2.1

struct Component {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0, pred = 0;
    int bw = 0, bh = 0;                 // blocks per row / column of the plane (padded to whole MCUs)
    std::vector<int16_t> coef;          // bw*bh blocks x 64 quantised coefficients in natural order (scans accumulate here)
    std::vector<uint8_t> plane;         // bw*8 x bh*8 samples, after the inverse DCT
};

struct IdctTable {
    double c[8][8];                     // c[x][u] = C(u)/2 * cos((2x+1) u pi / 16)
    IdctTable() {
        for (int x = 0; x < 8; x++)
            for (int u = 0; u < 8; u++) c[x][u] = (u == 0 ? std::sqrt(0.5) : 1.0) * 0.5 * std::cos((2 * x + 1) * u * 3.14159265358979323846 / 16.0);
    }
};

void idct_block(const int coef[64], uint8_t* dst, int stride) {
    static const IdctTable T;
    double tmp[64];
    for (int v = 0; v < 8; v++)          // rows: over u
        for (int x = 0; x < 8; x++) {
            double s = 0.0;
            for (int u = 0; u < 8; u++) s += T.c[x][u] * coef[v * 8 + u];
            tmp[v * 8 + x] = s;
        }
    for (int x = 0; x < 8; x++)          // columns: over v
        for (int y = 0; y < 8; y++) {
            double s = 0.0;
            for (int v = 0; v < 8; v++) s += T.c[y][v] * tmp[v * 8 + x];
            const long r = std::lround(s + 128.0);
            dst[y * stride + x] = (uint8_t)(r < 0 ? 0 : (r > 255 ? 255 : r));
        }
}

uint16_t be16(const uint8_t* p) { return (uint16_t)((p[0] << 8) | p[1]); }

// Chroma upsampling of a component plane (cw x ch valid samples, row stride `stride`) to the full W x H: the triangle
// ("fancy") filters of the IJG library for 2:1 horizontal and 2:1 x 2:1 — what libjpeg(-turbo) and the Rust decoders apply by
// default — and sample replication for every other ratio.
void upsample(const Component& c, int hmax, int vmax, uint32_t W, uint32_t H, std::vector<uint8_t>& out) {
    out.resize((size_t)W * H);
    const int stride = c.bw * 8;
    const int cw = (int)((W * (uint32_t)c.h + hmax - 1) / hmax), ch = (int)((H * (uint32_t)c.v + vmax - 1) / vmax);
    const uint8_t* p = c.plane.data();
    const bool h2 = hmax == 2 * c.h, v1 = vmax == c.v, v2 = vmax == 2 * c.v;
    if (h2 && v1 && cw >= 2) {                                           // h2v1_fancy_upsample
        for (uint32_t y = 0; y < H; y++) {
            const uint8_t* in = p + (size_t)y * stride;
            uint8_t* o = out.data() + (size_t)y * W;
            for (int i = 0; i < cw; i++) {
                const int v = in[i] * 3, l = i > 0 ? in[i - 1] : -1, r = i + 1 < cw ? in[i + 1] : -1;
                const int a = l < 0 ? in[i] : (v + l + 1) >> 2, b = r < 0 ? in[i] : (v + r + 2) >> 2;
                if ((uint32_t)(2 * i) < W) o[2 * i] = (uint8_t)a;
                if ((uint32_t)(2 * i + 1) < W) o[2 * i + 1] = (uint8_t)b;
            }
        }
        return;
    }
    if (h2 && v2 && cw >= 2) {                                           // h2v2_fancy_upsample: 3/4 nearer row + 1/4 farther row, then 3:1 across
        std::vector<int> sum((size_t)cw);
        for (uint32_t y = 0; y < H; y++) {
            const int r = (int)(y >> 1);
            int far = (y & 1u) ? r + 1 : r - 1;
            far = far < 0 ? 0 : (far >= ch ? ch - 1 : far);
            const uint8_t* in0 = p + (size_t)r * stride;
            const uint8_t* in1 = p + (size_t)far * stride;
            for (int i = 0; i < cw; i++) sum[(size_t)i] = in0[i] * 3 + in1[i];
            uint8_t* o = out.data() + (size_t)y * W;
            for (int i = 0; i < cw; i++) {
                const int t = sum[(size_t)i];
                const int a = i > 0 ? (t * 3 + sum[(size_t)i - 1] + 8) >> 4 : (t * 4 + 8) >> 4;
                const int b = i + 1 < cw ? (t * 3 + sum[(size_t)i + 1] + 7) >> 4 : (t * 4 + 7) >> 4;
                if ((uint32_t)(2 * i) < W) o[2 * i] = (uint8_t)a;
                if ((uint32_t)(2 * i + 1) < W) o[2 * i + 1] = (uint8_t)b;
            }
        }
        return;
    }
    for (uint32_t y = 0; y < H; y++)
        for (uint32_t x = 0; x < W; x++) out[(size_t)y * W + x] = p[(size_t)(y * (uint32_t)c.v / (uint32_t)vmax) * stride + x * (uint32_t)c.h / (uint32_t)hmax];
}

}  // namespace

bool decode_jpeg(const uint8_t* d, size_t n, uint32_t& width, uint32_t& height, uint32_t& channels, std::vector<uint8_t>& pixels, std::string& err) {
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) { err = "image: not a JPEG"; return false; }
    uint16_t qt[4][64]; bool have_qt[4] = {false, false, false, false};
    Huffman dc[4], ac[4];
    std::vector<Component> comps;
    int hmax = 1, vmax = 1, restart_interval = 0, adobe_transform = -1;
    bool have_frame = false, have_scan = false, progressive = false;
    size_t pos = 2;
    while (pos + 4 <= n) {
        if (d[pos] != 0xFF) { err = "image: JPEG marker expected"; return false; }
        while (pos < n && d[pos] == 0xFF) pos++;                       // fill bytes
        if (pos >= n) break;
        const uint8_t m = d[pos++];
        if (m == 0xD9) break;                                          // EOI
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;           // TEM / stray RSTn: no length
        if (pos + 2 > n) { err = "image: truncated JPEG"; return false; }
        const size_t len = be16(d + pos);
        if (len < 2 || pos + len > n) { err = "image: truncated JPEG segment"; return false; }
        const uint8_t* seg = d + pos + 2;
        const size_t sl = len - 2;
        if (m == 0xDB) {                                               // DQT
            size_t i = 0;
            while (i < sl) {
                const int pq = seg[i] >> 4, tq = seg[i] & 15;
                i++;
                if (tq > 3 || i + (pq ? 128u : 64u) > sl) { err = "image: bad JPEG quantisation table"; return false; }
                for (int k = 0; k < 64; k++) { qt[tq][kZigzag[k]] = pq ? be16(seg + i + 2 * k) : seg[i + k]; }
                i += pq ? 128 : 64;
                have_qt[tq] = true;
            }
        } else if (m == 0xC4) {                                        // DHT
            size_t i = 0;
            while (i + 17 <= sl) {
                const int tc = seg[i] >> 4, th = seg[i] & 15;
                int total = 0;
                for (int k = 0; k < 16; k++) total += seg[i + 1 + k];
                if (tc > 1 || th > 3 || total > 256 || i + 17 + (size_t)total > sl) { err = "image: bad JPEG Huffman table"; return false; }
                (tc ? ac[th] : dc[th]).build(seg + i + 1, seg + i + 17);
                i += 17 + (size_t)total;
            }
        } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {              // SOF0 baseline / SOF1 extended sequential / SOF2 progressive (Huffman)
            progressive = m == 0xC2;
            if (have_frame) { err = "image: JPEG with several frames"; return false; }
            if (sl < 6 || seg[0] != 8) { err = "image: only 8-bit JPEG samples are supported"; return false; }
            height = be16(seg + 1); width = be16(seg + 3);
            const int nc = seg[5];
            if (width == 0 || height == 0) { err = "image: JPEG with an empty extent (DNL is not supported)"; return false; }
            if (nc != 1 && nc != 3) { err = "image: JPEG with " + std::to_string(nc) + " components is not supported (greyscale and YCbCr are)"; return false; }
            if (sl < 6 + 3u * nc) { err = "image: truncated JPEG frame header"; return false; }
            comps.resize((size_t)nc);
            for (int c = 0; c < nc; c++) {
                comps[c].id = seg[6 + 3 * c]; comps[c].h = seg[7 + 3 * c] >> 4; comps[c].v = seg[7 + 3 * c] & 15; comps[c].tq = seg[8 + 3 * c];
                if (comps[c].h < 1 || comps[c].h > 4 || comps[c].v < 1 || comps[c].v > 4 || comps[c].tq > 3) { err = "image: bad JPEG sampling factors"; return false; }
                hmax = comps[c].h > hmax ? comps[c].h : hmax; vmax = comps[c].v > vmax ? comps[c].v : vmax;
            }
            const int mcux = (int)((width + 8u * hmax - 1) / (8u * hmax)), mcuy = (int)((height + 8u * vmax - 1) / (8u * vmax));
            for (auto& c : comps) {
                c.bw = mcux * c.h; c.bh = mcuy * c.v;
                if ((uint64_t)c.bw * c.bh > (1u << 24)) { err = "image: JPEG too large"; return false; }
                c.coef.assign((size_t)c.bw * c.bh * 64, 0);
            }
            have_frame = true;
        }
        else if (m == 0xC3 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) { err = "image: lossless / arithmetic-coded / hierarchical JPEG is not supported"; return false; }
        else if (m == 0xDD) { if (sl >= 2) restart_interval = be16(seg); }
        else if (m == 0xEE) { if (sl >= 12 && !memcmp(seg, "Adobe", 5)) adobe_transform = seg[11]; }
        else if (m == 0xDA) {                                          // SOS + entropy-coded data
            if (!have_frame) { err = "image: JPEG scan before the frame header"; return false; }
            if (sl < 1) { err = "image: truncated JPEG scan header"; return false; }
            const int ns = seg[0];
            if (ns < 1 || ns > (int)comps.size() || sl < 1 + 2u * ns + 3u) { err = "image: bad JPEG scan header"; return false; }
            const int Ss = seg[1 + 2 * ns], Se = seg[2 + 2 * ns], Ah = seg[3 + 2 * ns] >> 4, Al = seg[3 + 2 * ns] & 15;
            if (!progressive) { if (Ss != 0 || Se != 63 || Ah != 0 || Al != 0) { err = "image: bad spectral selection in a sequential JPEG scan"; return false; } }
            else if (Ss > Se || Se > 63 || Al > 13 || Ah > 13 || (Ss == 0 && Se != 0) || (Ss > 0 && ns != 1)) { err = "image: bad progressive JPEG scan parameters"; return false; }
            const bool dc_scan = Ss == 0, need_ac = !progressive || Ss > 0, need_dc = dc_scan && Ah == 0;
            std::vector<Component*> sc;
            for (int k = 0; k < ns; k++) {
                Component* c = nullptr;
                for (auto& cc : comps) if (cc.id == seg[1 + 2 * k]) c = &cc;
                if (!c) { err = "image: JPEG scan names an unknown component"; return false; }
                c->td = seg[2 + 2 * k] >> 4; c->ta = seg[2 + 2 * k] & 15;
                if (c->td > 3 || c->ta > 3 || (need_dc && !dc[c->td].present) || (need_ac && !ac[c->ta].present)) { err = "image: JPEG scan uses a table that was not defined"; return false; }
                c->pred = 0;
                sc.push_back(c);
            }
            BitReader br{d, n, pos + len};
            const int mcux = (int)((width + 8u * hmax - 1) / (8u * hmax)), mcuy = (int)((height + 8u * vmax - 1) / (8u * vmax));
            // a single-component scan is NOT interleaved: its units are the component's own 8x8 blocks covering its extent (A.2.2)
            const bool single = ns == 1;
            const int ux = single ? (int)(((width * sc[0]->h + hmax - 1) / hmax + 7) / 8) : mcux;
            const int uy = single ? (int)(((height * sc[0]->v + vmax - 1) / vmax + 7) / 8) : mcuy;
            int since_restart = 0, expected_rst = 0, eobrun = 0;
            const int p1 = 1 << Al, m1 = -(1 << Al);
            bool ok = true;
            for (int my = 0; my < uy && ok; my++)
                for (int mx = 0; mx < ux && ok; mx++) {
                    if (restart_interval && since_restart == restart_interval) {          // RSTn: byte-align, reset predictors (F.2.1.3.1)
                        br.align();
                        size_t p = br.pos;
                        while (p + 1 < n && !(d[p] == 0xFF && d[p + 1] >= 0xD0 && d[p + 1] <= 0xD7)) p++;
                        if (p + 1 >= n || d[p + 1] != 0xD0 + expected_rst) { err = "image: JPEG restart marker missing"; return false; }
                        br.pos = p + 2; br.hit_marker = false; br.cnt = 0;
                        expected_rst = (expected_rst + 1) & 7;
                        for (auto* c : sc) c->pred = 0;
                        eobrun = 0;
                        since_restart = 0;
                    }
                    for (auto* c : sc) {
                        const int nh = single ? 1 : c->h, nv = single ? 1 : c->v;
                        for (int by = 0; by < nv && ok; by++)
                            for (int bx = 0; bx < nh && ok; bx++) {
                                const int bxx = (single ? mx : mx * c->h + bx), byy = (single ? my : my * c->v + by);
                                int16_t scratch[64];
                                int16_t* blk = (bxx < c->bw && byy < c->bh) ? c->coef.data() + ((size_t)byy * c->bw + bxx) * 64 : scratch;
                                if (blk == scratch) memset(scratch, 0, sizeof(scratch));
                                if (dc_scan) {
                                    if (Ah == 0) {                                        // DC, first pass (F.2.2.1 / G.1.2.1)
                                        const int sz = decode_symbol(br, dc[c->td], ok);
                                        if (!ok || sz > 11) { ok = false; break; }
                                        c->pred += extend(br.bits(sz), sz);
                                        if (c->pred < -32768 || c->pred > 32767) { ok = false; break; }   // an 8-bit image's DC fits 12 bits: corrupt data
                                        const int v = c->pred * p1;
                                        if (v < -32768 || v > 32767) { ok = false; break; }
                                        blk[0] = (int16_t)v;
                                    } else if (br.bit()) blk[0] = (int16_t)(blk[0] | p1);   // DC refinement (G.1.2.1)
                                }
                                if (!progressive) {                                       // sequential AC (F.2.2.2)
                                    for (int k = 1; k < 64;) {
                                        const int rs = decode_symbol(br, ac[c->ta], ok);
                                        if (!ok) break;
                                        const int r = rs >> 4, sz = rs & 15;
                                        if (sz == 0) { if (r == 15) { k += 16; continue; } break; }      // ZRL / EOB
                                        k += r;
                                        if (k > 63 || sz > 11) { ok = false; break; }                     // AC categories of 8-bit data are <= 10
                                        blk[kZigzag[k]] = (int16_t)extend(br.bits(sz), sz);
                                        k++;
                                    }
                                } else if (!dc_scan && Ah == 0) {                         // progressive AC, first pass (G.1.2.2)
                                    if (eobrun > 0) { eobrun--; continue; }
                                    for (int k = Ss; k <= Se; k++) {
                                        const int rs = decode_symbol(br, ac[c->ta], ok);
                                        if (!ok) break;
                                        const int r = rs >> 4, sz = rs & 15;
                                        if (sz) {
                                            k += r;
                                            if (k > Se || sz > 11) { ok = false; break; }
                                            const int v = extend(br.bits(sz), sz) * p1;
                                            if (v < -32768 || v > 32767) { ok = false; break; }
                                            blk[kZigzag[k]] = (int16_t)v;
                                        } else if (r == 15) k += 15;
                                        else { eobrun = (1 << r) + (r ? br.bits(r) : 0) - 1; break; }
                                    }
                                } else if (!dc_scan) {                                    // progressive AC, refinement (G.1.2.3)
                                    int k = Ss;
                                    if (eobrun == 0) {
                                        for (; k <= Se; k++) {
                                            const int rs = decode_symbol(br, ac[c->ta], ok);
                                            if (!ok) break;
                                            int r = rs >> 4, sv = rs & 15;
                                            if (sv) { if (sv != 1) { ok = false; break; } sv = br.bit() ? p1 : m1; }
                                            else if (r != 15) { eobrun = (1 << r) + (r ? br.bits(r) : 0); break; }
                                            do {                                          // pass over nonzero history (correction bits) and r zero coefficients
                                                int16_t& t = blk[kZigzag[k]];
                                                if (t != 0) { if (br.bit() && (t & p1) == 0) t = (int16_t)(t + (t >= 0 ? p1 : m1)); }
                                                else if (--r < 0) break;
                                                k++;
                                            } while (k <= Se);
                                            if (sv && k <= Se) blk[kZigzag[k]] = (int16_t)sv;
                                        }
                                    }
                                    if (ok && eobrun > 0) {
                                        for (; k <= Se; k++) {
                                            int16_t& t = blk[kZigzag[k]];
                                            if (t != 0 && br.bit() && (t & p1) == 0) t = (int16_t)(t + (t >= 0 ? p1 : m1));
                                        }
                                        eobrun--;
                                    }
                                }
                            }
                        if (!ok) break;
                    }
                    since_restart++;
                }
            if (!ok) { err = "image: corrupt JPEG entropy-coded data"; return false; }
            if (br.ran_out) { err = "image: truncated JPEG (entropy-coded data ends before the image does)"; return false; }
            have_scan = true;
            // continue after the entropy-coded segment: at the marker the reader stopped at
            size_t p = br.pos;
            while (p + 1 < n && !(d[p] == 0xFF && d[p + 1] != 0x00 && !(d[p + 1] >= 0xD0 && d[p + 1] <= 0xD7))) p++;
            pos = p;
            continue;
        }
        pos += len;
    }
    if (!have_frame || !have_scan) { err = "image: JPEG without image data"; return false; }
    for (auto& c : comps) {                                            // dequantise + inverse DCT of every block
        if (!have_qt[c.tq]) { err = "image: JPEG component uses a quantisation table that was not defined"; return false; }
        c.plane.assign((size_t)c.bw * 8 * c.bh * 8, 128);
        for (int by = 0; by < c.bh; by++)
            for (int bx = 0; bx < c.bw; bx++) {
                const int16_t* q = c.coef.data() + ((size_t)by * c.bw + bx) * 64;
                int coef[64];
                for (int k = 0; k < 64; k++) coef[k] = (int)q[k] * (int)qt[c.tq][k];
                idct_block(coef, c.plane.data() + ((size_t)by * 8 * c.bw + bx) * 8, c.bw * 8);
            }
        std::vector<int16_t>().swap(c.coef);
    }
    channels = (uint32_t)comps.size();
    pixels.resize((size_t)width * height * channels);
    const bool ycc = comps.size() == 3 && adobe_transform != 0;        // JFIF / Adobe transform 1: YCbCr; Adobe transform 0: RGB as is
    std::vector<uint8_t> full[3];
    for (size_t c = 0; c < comps.size(); c++) upsample(comps[c], hmax, vmax, width, height, full[c]);
    for (uint32_t y = 0; y < height; y++)
        for (uint32_t x = 0; x < width; x++) {
            int s[3] = {0, 0, 0};
            for (size_t c = 0; c < comps.size(); c++) s[c] = full[c][(size_t)y * width + x];
            uint8_t* q = &pixels[((size_t)y * width + x) * channels];
            if (comps.size() == 1) q[0] = (uint8_t)s[0];
            else if (!ycc) { q[0] = (uint8_t)s[0]; q[1] = (uint8_t)s[1]; q[2] = (uint8_t)s[2]; }
            else {
                const double Y = s[0], cb = s[1] - 128.0, cr = s[2] - 128.0;
                const long r = std::lround(Y + 1.402 * cr), g = std::lround(Y - 0.344136 * cb - 0.714136 * cr), b = std::lround(Y + 1.772 * cb);
                q[0] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r); q[1] = (uint8_t)(g < 0 ? 0 : g > 255 ? 255 : g); q[2] = (uint8_t)(b < 0 ? 0 : b > 255 ? 255 : b);
            }
        }
    return true;
}

}  // namespace srh
