// glTF 2.0 ingest for the ray-tracing path: what `Gltf::new` + `create_default_scene`
// (src/vulkan_abstraction/gltf/mod.rs:57-373) and the CPU side of `Scene::load_into_gpu` (src/scene.rs:52-176)
// produce — unique BLAS inputs (96-byte vertices, u32 indices, material, local emissive triangles), the
// (blas index, world transform) instance list, textures, samplers and decoded images.
//
// The reference parses with the `gltf` crate (1.4.1, Cargo.lock) and decodes images with `image` (0.25.10);
// neither is vendored under /root/reference, so the container format, accessor rules, node-transform and
// material-default rules are restated from the glTF 2.0 specification, anchored on the reference's call
// sites. Host-only code: no HIP here.
#include <zlib.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "host.h"

namespace {

// ---- minimal JSON DOM ------------------------------------------------------------------------
struct Json {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;
    const Json* get(const char* key) const {
        if (kind != Obj) return nullptr;
        for (const auto& kv : obj) if (kv.first == key) return &kv.second;
        return nullptr;
    }
    bool has(const char* key) const { return get(key) != nullptr; }
    double number(const char* key, double dflt) const { const Json* j = get(key); return (j && j->kind == Num) ? j->num : dflt; }
    long index(const char* key) const {      // non-negative integer below 2^31, else -1
        const Json* j = get(key);
        return (j && j->kind == Num && j->num >= 0.0 && j->num < 2147483648.0) ? (long)j->num : -1;
    }
    size_t size_value(const char* key) const {   // byte offsets / counts: clamped to [0, 2^48)
        const Json* j = get(key);
        if (!j || j->kind != Num || !(j->num >= 0.0)) return 0;
        return j->num < 281474976710656.0 ? (size_t)j->num : (size_t)281474976710655ull;
    }
    size_t size() const { return kind == Arr ? arr.size() : 0; }
};

struct JsonParser {
    const char* p; const char* end; std::string err;
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\r' || *p == '\t')) p++; }
    bool fail(const char* m) { if (err.empty()) err = m; return false; }
    bool parse_string(std::string& out) {
        if (p >= end || *p != '"') return fail("json: expected string");
        p++;
        while (p < end && *p != '"') {
            if (*p == '\\') {
                if (++p >= end) return fail("json: bad escape");
                switch (*p) {
                    case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break; case 'f': out += '\f'; break;
                    case 'u': {
                        if (end - p < 5) return fail("json: bad \\u escape");
                        unsigned cp = 0;
                        for (int i = 1; i <= 4; i++) {
                            char c = p[i]; cp <<= 4;
                            if (c >= '0' && c <= '9') cp |= c - '0'; else if (c >= 'a' && c <= 'f') cp |= c - 'a' + 10;
                            else if (c >= 'A' && c <= 'F') cp |= c - 'A' + 10; else return fail("json: bad \\u escape");
                        }
                        p += 4;
                        if (cp < 0x80) out += (char)cp;
                        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
                        else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
                        break;
                    }
                    default: out += *p;
                }
                p++;
            } else out += *p++;
        }
        if (p >= end) return fail("json: unterminated string");
        p++;
        return true;
    }
    bool parse(Json& out, int depth = 0) {
        if (depth > 64) return fail("json: nesting too deep");
        ws();
        if (p >= end) return fail("json: unexpected end");
        if (*p == '{') {
            out.kind = Json::Obj; p++; ws();
            if (p < end && *p == '}') { p++; return true; }
            for (;;) {
                ws();
                std::string k;
                if (!parse_string(k)) return false;
                ws();
                if (p >= end || *p != ':') return fail("json: expected ':'");
                p++;
                out.obj.emplace_back(std::move(k), Json());
                if (!parse(out.obj.back().second, depth + 1)) return false;
                ws();
                if (p < end && *p == ',') { p++; continue; }
                if (p < end && *p == '}') { p++; return true; }
                return fail("json: expected ',' or '}'");
            }
        }
        if (*p == '[') {
            out.kind = Json::Arr; p++; ws();
            if (p < end && *p == ']') { p++; return true; }
            for (;;) {
                out.arr.emplace_back();
                if (!parse(out.arr.back(), depth + 1)) return false;
                ws();
                if (p < end && *p == ',') { p++; continue; }
                if (p < end && *p == ']') { p++; return true; }
                return fail("json: expected ',' or ']'");
            }
        }
        if (*p == '"') { out.kind = Json::Str; return parse_string(out.str); }
        if (end - p >= 4 && !strncmp(p, "true", 4)) { out.kind = Json::Bool; out.b = true; p += 4; return true; }
        if (end - p >= 5 && !strncmp(p, "false", 5)) { out.kind = Json::Bool; out.b = false; p += 5; return true; }
        if (end - p >= 4 && !strncmp(p, "null", 4)) { out.kind = Json::Null; p += 4; return true; }
        const char* q = p;
        while (q < end && (*q == '-' || *q == '+' || *q == '.' || *q == 'e' || *q == 'E' || (*q >= '0' && *q <= '9'))) q++;
        if (q == p) return fail("json: unexpected character");
        std::string t(p, q);
        char* e = nullptr;
        out.num = strtod(t.c_str(), &e);
        if (e == t.c_str()) return fail("json: bad number");
        out.kind = Json::Num; p = q;
        return true;
    }
};

// ---- files, base64, PNG ----------------------------------------------------------------------
bool read_file(const std::string& path, std::vector<uint8_t>& out) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (n < 0) { fclose(f); return false; }
    out.resize((size_t)n);
    size_t got = n ? fread(out.data(), 1, (size_t)n, f) : 0;
    fclose(f);
    return got == (size_t)n;
}

bool base64_decode(const char* s, size_t n, std::vector<uint8_t>& out) {
    auto val = [](char c) -> int {
        if (c >= 'A' && c <= 'Z') return c - 'A'; if (c >= 'a' && c <= 'z') return c - 'a' + 26;
        if (c >= '0' && c <= '9') return c - '0' + 52; if (c == '+' || c == '-') return 62; if (c == '/' || c == '_') return 63;
        return -1;
    };
    uint32_t acc = 0; int bits = 0;
    for (size_t i = 0; i < n; i++) {
        if (s[i] == '=' || s[i] == '\n' || s[i] == '\r') continue;
        int v = val(s[i]);
        if (v < 0) return false;
        acc = (acc << 6) | (uint32_t)v; bits += 6;
        if (bits >= 8) { bits -= 8; out.push_back((uint8_t)(acc >> bits)); }
    }
    return true;
}

struct DecodedImage { uint32_t w = 0, h = 0, channels = 0; std::vector<uint8_t> pixels; };

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

// PNG (ISO/IEC 15948) -> 8-bit pixels the way image-rs hands them to the gltf crate: greyscale = R8, grey+alpha =
// R8G8, truecolour = R8G8B8, truecolour+alpha = R8G8B8A8, palette = expanded to RGB (RGBA with tRNS); bit depths
// 1/2/4 are scaled to 8; a tRNS chunk of a greyscale / truecolour image becomes an alpha channel (the png crate's EXPAND
// transformation: alpha 0 where the pixel equals the chunk's colour at the file's bit depth, else opaque). 16-bit PNGs map to
// R16* formats, which Image::new_from_data does not handle (todo!(), image/mod.rs:102-107) -> rejected for glTF textures;
// `allow16` (the to_rgba8 path of lib.rs:281-283) narrows every 16-bit sample v to (v + 128) / 257, image-rs 0.25's
// u16 -> u8 conversion. Adam7 interlacing is not supported.
bool decode_png(const uint8_t* d, size_t n, DecodedImage& out, std::string& err, bool allow16 = false) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (n < 8 || memcmp(d, sig, 8) != 0) { err = "image: not a PNG"; return false; }
    size_t pos = 8;
    uint32_t w = 0, h = 0; int depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte, trns;
    while (pos + 12 <= n) {
        uint32_t len = be32(d + pos);
        const uint8_t* type = d + pos + 4;
        if (pos + 12 + (size_t)len > n) { err = "image: truncated PNG chunk"; return false; }
        const uint8_t* body = d + pos + 8;
        if (!memcmp(type, "IHDR", 4) && len >= 13) { w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12]; }
        else if (!memcmp(type, "PLTE", 4)) plte.assign(body, body + len);
        else if (!memcmp(type, "tRNS", 4)) trns.assign(body, body + len);
        else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!memcmp(type, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    if (w == 0 || h == 0 || ctype < 0) { err = "image: PNG without IHDR"; return false; }
    if (w > 32768 || h > 32768) { err = "image: PNG larger than 32768 pixels on a side"; return false; }
    if (interlace) { err = "image: interlaced PNG is not supported"; return false; }
    if (depth == 16 && !allow16) { err = "image: 16-bit PNG maps to an R16 format the reference's Image::new_from_data does not handle (image/mod.rs:102-107)"; return false; }
    int samples = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (!samples || (depth != 16 && depth != 8 && depth != 4 && depth != 2 && depth != 1) || (depth < 8 && ctype != 0 && ctype != 3) || (depth == 16 && ctype == 3)) { err = "image: unsupported PNG colour type / bit depth"; return false; }
    const size_t bpp_bits = (size_t)samples * depth, stride = ((size_t)w * bpp_bits + 7) / 8, bpp = (bpp_bits + 7) / 8;
    // deflate expands at most ~1032:1: refuse a header that promises more than the IDAT bytes can hold before allocating for it
    if ((stride + 1) * (size_t)h > idat.size() * 1032 + 65536) { err = "image: PNG IDAT data too short for the extent in its header"; return false; }
    std::vector<uint8_t> raw((stride + 1) * h);
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) { err = "image: PNG inflate failed"; return false; }
    std::vector<uint8_t> img(stride * h), zero(stride, 0);
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t* src = raw.data() + (stride + 1) * y;
        const uint8_t* up = y ? img.data() + stride * (y - 1) : zero.data();
        uint8_t* cur = img.data() + stride * y;
        const int ft = src[0];
        for (size_t x = 0; x < stride; x++) {
            const int a = x >= bpp ? cur[x - bpp] : 0, b = up[x], c = x >= bpp ? up[x - bpp] : 0;
            int pred = 0;
            if (ft == 1) pred = a; else if (ft == 2) pred = b; else if (ft == 3) pred = (a + b) >> 1;
            else if (ft == 4) { int pp = a + b - c, pa = abs(pp - a), pb = abs(pp - b), pc = abs(pp - c); pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c); }
            else if (ft != 0) { err = "image: bad PNG filter type"; return false; }
            cur[x] = (uint8_t)(src[1 + x] + pred);
        }
    }
    auto sample = [&](uint32_t y, size_t i) -> uint32_t {   // i-th sample of row y at `depth` bits
        const uint8_t* row = img.data() + stride * y;
        if (depth == 8) return row[i];
        if (depth == 16) return ((uint32_t)row[2 * i] << 8) | row[2 * i + 1];
        const size_t bit = i * depth;
        return (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1);
    };
    out.w = w; out.h = h;
    if (ctype == 3) {
        const bool alpha = !trns.empty();
        out.channels = alpha ? 4 : 3;
        out.pixels.resize((size_t)w * h * out.channels);
        for (uint32_t y = 0; y < h; y++)
            for (uint32_t x = 0; x < w; x++) {
                const uint32_t k = sample(y, x);
                if ((size_t)k * 3 + 2 >= plte.size()) { err = "image: PNG palette index out of range"; return false; }
                uint8_t* q = &out.pixels[((size_t)y * w + x) * out.channels];
                q[0] = plte[k * 3]; q[1] = plte[k * 3 + 1]; q[2] = plte[k * 3 + 2];
                if (alpha) q[3] = k < trns.size() ? trns[k] : 255;
            }
        return true;
    }
    // tRNS of a greyscale (2 bytes) or truecolour (6 bytes) image: one transparent colour, 16 bits per sample in the chunk
    const bool keyed = (ctype == 0 && trns.size() >= 2) || (ctype == 2 && trns.size() >= 6);
    uint32_t key[3] = {0, 0, 0};
    for (int c = 0; keyed && c < samples; c++) key[c] = (((uint32_t)trns[2 * c] << 8) | trns[2 * c + 1]) & ((1u << depth) - 1);
    out.channels = (uint32_t)samples + (keyed ? 1u : 0u);
    out.pixels.resize((size_t)w * h * out.channels);
    const uint32_t scale = depth >= 8 ? 1 : 255u / ((1u << depth) - 1);
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) {
            uint8_t* q = &out.pixels[((size_t)y * w + x) * out.channels];
            bool is_key = keyed;
            for (int c = 0; c < samples; c++) {
                const uint32_t v = sample(y, (size_t)x * samples + c);
                if (keyed && v != key[c]) is_key = false;
                q[c] = depth == 16 ? (uint8_t)((v + 128u) / 257u) : (uint8_t)(v * scale);
            }
            if (keyed) q[samples] = is_key ? 0 : 255;
        }
    return true;
}

// ---- the loader ---------------------------------------------------------------------------------
struct Mat4 { float m[16]; };   // row-major
Mat4 identity() { Mat4 r; memset(&r, 0, sizeof(r)); r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.0f; return r; }
Mat4 mul(const Mat4& a, const Mat4& b) {      // parent_transform * node matrix (gltf/mod.rs:173), fp32, k = 0..3 in order
    Mat4 r;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            r.m[i * 4 + j] = ((a.m[i * 4] * b.m[j] + a.m[i * 4 + 1] * b.m[4 + j]) + a.m[i * 4 + 2] * b.m[8 + j]) + a.m[i * 4 + 3] * b.m[12 + j];
    return r;
}

}  // namespace

namespace srh {

struct GltfBlas {
    std::vector<SrVertex> vertices;
    std::vector<uint32_t> indices;
    SrMaterial material;      // *_image = glTF TEXTURE index (or SR_NULL_TEXTURE), *_sampler = SR_NULL_TEXTURE: unresolved
    std::vector<SrEmissiveTriangle> emissive;
};
struct GltfTexture { int32_t sampler; uint32_t source; };
struct GltfScene {
    std::vector<GltfBlas> blases;
    std::vector<std::pair<uint32_t, SrTransform>> instances;   // (blas index, world transform), scene.rs:31-33
    std::vector<GltfTexture> textures;
    std::vector<SrSamplerDesc> samplers;
    std::vector<DecodedImage> images;
};

struct GltfLoader {
    Json doc;
    std::string dir;
    std::vector<std::vector<uint8_t>> buffers;
    std::vector<uint8_t> glb_bin;
    bool has_glb_bin = false;
    std::string err;
    int err_code = SR_ERR_INVALID_ARG;
    GltfScene out;
    struct PrimData { std::vector<SrVertex> vertices; std::vector<uint32_t> indices; };
    typedef std::pair<long, long> PrimKey;                       // (POSITION accessor, indices accessor | i), gltf/mod.rs:200-212
    std::map<PrimKey, PrimData> primitive_data_map;
    std::map<PrimKey, uint32_t> primitives_blas_index;           // scene.rs:56

    bool fail(const std::string& m, int code = SR_ERR_INVALID_ARG) { if (err.empty()) { err = "gltf: " + m; err_code = code; } return false; }
    const Json* element(const char* array, long i) const {
        const Json* a = doc.get(array);
        return (a && a->kind == Json::Arr && i >= 0 && (size_t)i < a->arr.size()) ? &a->arr[i] : nullptr;
    }

    bool resolve_uri(const std::string& uri, std::vector<uint8_t>& bytes) {
        if (uri.compare(0, 5, "data:") == 0) {
            size_t comma = uri.find(',');
            if (comma == std::string::npos || uri.find(";base64") == std::string::npos) return fail("unsupported data: URI");
            if (!base64_decode(uri.c_str() + comma + 1, uri.size() - comma - 1, bytes)) return fail("bad base64 payload");
            return true;
        }
        std::string path;
        for (size_t i = 0; i < uri.size(); i++) {       // percent-decoding of relative paths
            if (uri[i] == '%' && i + 2 < uri.size()) { path += (char)strtol(uri.substr(i + 1, 2).c_str(), nullptr, 16); i += 2; }
            else path += uri[i];
        }
        if (!read_file(dir + path, bytes)) return fail("cannot read '" + dir + path + "'");
        return true;
    }

    bool open(const std::string& path) {
        std::vector<uint8_t> file;
        if (!read_file(path, file)) return fail("cannot read '" + path + "'");
        size_t slash = path.find_last_of('/');
        dir = slash == std::string::npos ? "" : path.substr(0, slash + 1);
        const char* json_begin; const char* json_end;
        if (file.size() >= 12 && !memcmp(file.data(), "glTF", 4)) {
            uint32_t version, length;
            memcpy(&version, file.data() + 4, 4); memcpy(&length, file.data() + 8, 4);
            if (version != 2) return fail("unsupported GLB container version");
            if (length > file.size()) return fail("GLB length exceeds the file");
            size_t pos = 12; json_begin = json_end = nullptr;
            while (pos + 8 <= length) {
                uint32_t clen, ctype;
                memcpy(&clen, file.data() + pos, 4); memcpy(&ctype, file.data() + pos + 4, 4);
                if (pos + 8 + (size_t)clen > length) return fail("GLB chunk exceeds the file");
                if (ctype == 0x4E4F534Au && !json_begin) { json_begin = (const char*)file.data() + pos + 8; json_end = json_begin + clen; }
                else if (ctype == 0x004E4942u && !has_glb_bin) { glb_bin.assign(file.data() + pos + 8, file.data() + pos + 8 + clen); has_glb_bin = true; }
                pos += 8 + (size_t)clen;
            }
            if (!json_begin) return fail("GLB without a JSON chunk");
        } else { json_begin = (const char*)file.data(); json_end = json_begin + file.size(); }
        JsonParser jp{json_begin, json_end, ""};
        if (!jp.parse(doc) || doc.kind != Json::Obj) return fail(jp.err.empty() ? "document is not a JSON object" : jp.err);
        const Json* bufs = doc.get("buffers");
        for (size_t i = 0; bufs && i < bufs->size(); i++) {
            buffers.emplace_back();
            const Json* uri = bufs->arr[i].get("uri");
            if (uri && uri->kind == Json::Str) { if (!resolve_uri(uri->str, buffers.back())) return false; }
            else if (i == 0 && has_glb_bin) buffers.back() = glb_bin;
            else return fail("buffer without uri outside a GLB");
            if (buffers.back().size() < bufs->arr[i].size_value("byteLength")) return fail("buffer shorter than its byteLength");
        }
        return true;
    }

    // Accessor -> rows of `want_comps` floats (`into_f32`: normalized u8/u16/i8/i16 are divided by their max) or u32.
    struct View { const uint8_t* base = nullptr; size_t stride = 0, count = 0; int comp_type = 0, comps = 0; bool normalized = false; };
    // Byte range [offset, offset + need) of a bufferView (+ extra offset); false when it leaves the buffer.
    bool view_bytes(long view_index, size_t extra_offset, size_t need, const uint8_t** out) {
        const Json* bv = element("bufferViews", view_index);
        if (!bv) return fail("accessor without bufferView");
        const long bi = bv->index("buffer");
        if (bi < 0 || (size_t)bi >= buffers.size()) return fail("bufferView refers to a missing buffer");
        const size_t off = bv->size_value("byteOffset") + extra_offset;
        if (off > buffers[bi].size() || need > buffers[bi].size() - off) return fail("accessor exceeds its buffer");
        *out = buffers[bi].data() + off;
        return true;
    }
    std::deque<std::vector<uint8_t>> sparse_storage;      // materialised sparse accessors (Views point into these)
    bool accessor_view(long index, View& v) {
        const Json* a = element("accessors", index);
        if (!a) return fail("accessor index out of range");
        const Json* type = a->get("type");
        const std::string t = type && type->kind == Json::Str ? type->str : "";
        v.comps = t == "SCALAR" ? 1 : t == "VEC2" ? 2 : t == "VEC3" ? 3 : t == "VEC4" ? 4 : t == "MAT4" ? 16 : 0;
        v.comp_type = (int)a->number("componentType", 0);
        const size_t csz = (v.comp_type == 5120 || v.comp_type == 5121) ? 1 : (v.comp_type == 5122 || v.comp_type == 5123) ? 2 : (v.comp_type == 5125 || v.comp_type == 5126) ? 4 : 0;
        if (!v.comps || !csz) return fail("accessor with unsupported type / componentType");
        v.count = a->size_value("count");
        const Json* nj = a->get("normalized");
        v.normalized = nj && nj->kind == Json::Bool && nj->b;
        const size_t elem = csz * v.comps;
        const Json* sparse = a->get("sparse");
        const bool has_view = a->index("bufferView") >= 0;
        if (!has_view && !(sparse && sparse->kind == Json::Obj)) return fail("accessor without bufferView");
        if (v.count > (1ull << 32)) return fail("accessor exceeds its buffer");
        if (has_view) {
            const Json* bv = element("bufferViews", a->index("bufferView"));
            if (!bv) return fail("accessor without bufferView");
            v.stride = bv->size_value("byteStride");
            if (v.stride == 0) v.stride = elem;
            if (v.stride > 65536) return fail("accessor exceeds its buffer");
            if (!view_bytes(a->index("bufferView"), a->size_value("byteOffset"), v.count ? v.stride * (v.count - 1) + elem : 0, &v.base)) return false;
        }
        if (!sparse) return true;
        // Sparse accessor (glTF 2.0 §3.6.2.3, what the gltf crate's accessor::Iter resolves transparently for the readers of
        // gltf/mod.rs:57-67): the bufferView's elements — zeros without one — with `count` of them replaced by tightly packed
        // `values` at the strictly increasing positions `indices`. Materialised once, tightly packed.
        if (sparse->kind != Json::Obj) return fail("sparse accessor: malformed sparse object");
        const size_t n_sub = sparse->size_value("count");
        const Json* ji = sparse->get("indices");
        const Json* jv = sparse->get("values");
        if (!ji || ji->kind != Json::Obj || !jv || jv->kind != Json::Obj || n_sub == 0 || n_sub > v.count) return fail("sparse accessor: needs count (1..accessor count), indices and values");
        const int ict = (int)ji->number("componentType", 0);
        const size_t isz = ict == 5121 ? 1 : ict == 5123 ? 2 : ict == 5125 ? 4 : 0;
        if (!isz) return fail("sparse accessor: indices must be u8 / u16 / u32");
        const uint8_t *ip = nullptr, *vp = nullptr;
        if (!view_bytes(ji->index("bufferView"), ji->size_value("byteOffset"), n_sub * isz, &ip)) return false;
        if (!view_bytes(jv->index("bufferView"), jv->size_value("byteOffset"), n_sub * elem, &vp)) return false;
        sparse_storage.emplace_back(v.count * elem, (uint8_t)0);
        std::vector<uint8_t>& dst = sparse_storage.back();
        if (has_view) for (size_t i = 0; i < v.count; i++) memcpy(dst.data() + i * elem, v.base + i * v.stride, elem);
        long long last = -1;
        for (size_t k = 0; k < n_sub; k++) {
            uint32_t at = 0;
            if (isz == 1) at = ip[k];
            else if (isz == 2) { uint16_t u; memcpy(&u, ip + 2 * k, 2); at = u; }
            else memcpy(&at, ip + 4 * k, 4);
            if ((long long)at <= last || at >= v.count) return fail("sparse accessor: indices must be strictly increasing and inside the accessor");
            last = at;
            memcpy(dst.data() + (size_t)at * elem, vp + k * elem, elem);
        }
        v.base = dst.data();
        v.stride = elem;
        return true;
    }
    static float component_f32(const uint8_t* p, int ct, bool normalized) {
        switch (ct) {
            case 5126: { float f; memcpy(&f, p, 4); return f; }
            case 5121: return normalized ? (float)p[0] / 255.0f : (float)p[0];
            case 5123: { uint16_t u; memcpy(&u, p, 2); return normalized ? (float)u / 65535.0f : (float)u; }
            case 5120: { int8_t i; memcpy(&i, p, 1); return normalized ? fmaxf((float)i / 127.0f, -1.0f) : (float)i; }
            case 5122: { int16_t i; memcpy(&i, p, 2); return normalized ? fmaxf((float)i / 32767.0f, -1.0f) : (float)i; }
            default: { uint32_t u; memcpy(&u, p, 4); return (float)u; }
        }
    }
    bool read_floats(long accessor, int want_comps, std::vector<float>& outv, size_t* count) {
        View v;
        if (!accessor_view(accessor, v)) return false;
        if (v.comps != want_comps) return fail("accessor has the wrong number of components");
        const size_t csz = (v.comp_type == 5120 || v.comp_type == 5121) ? 1 : (v.comp_type == 5122 || v.comp_type == 5123) ? 2 : 4;
        outv.resize(v.count * want_comps);
        for (size_t i = 0; i < v.count; i++)
            for (int c = 0; c < want_comps; c++) outv[i * want_comps + c] = component_f32(v.base + i * v.stride + c * csz, v.comp_type, v.normalized || v.comp_type != 5126);
        *count = v.count;
        return true;
    }
    bool read_indices(long accessor, std::vector<uint32_t>& outv) {
        View v;
        if (!accessor_view(accessor, v)) return false;
        if (v.comps != 1 || (v.comp_type != 5121 && v.comp_type != 5123 && v.comp_type != 5125)) return fail("index accessor must be a u8/u16/u32 SCALAR");
        outv.resize(v.count);
        for (size_t i = 0; i < v.count; i++) {
            const uint8_t* p = v.base + i * v.stride;
            if (v.comp_type == 5121) outv[i] = p[0];
            else if (v.comp_type == 5123) { uint16_t u; memcpy(&u, p, 2); outv[i] = u; }
            else memcpy(&outv[i], p, 4);
        }
        return true;
    }

    // (texture index, texCoord set) of a textureInfo member: get_texture_indices! (gltf/mod.rs:28-35)
    static void texture_ref(const Json* owner, const char* name, uint32_t& tex, long& set) {
        tex = SR_NULL_TEXTURE; set = 0;
        const Json* t = owner ? owner->get(name) : nullptr;
        if (!t || t->kind != Json::Obj) return;
        const long i = t->index("index");
        if (i >= 0) { tex = (uint32_t)i; set = (long)t->number("texCoord", 0); }
    }

    struct MaterialInfo { SrMaterial m; long sets[5]; bool is_emissive; };
    MaterialInfo material_of(const Json& prim) {
        // gltf/mod.rs:214-262; a primitive without `material` gets the glTF default material (all factors 1, no emission)
        MaterialInfo r;
        memset(&r.m, 0, sizeof(r.m));
        const Json* mat = element("materials", prim.index("material"));
        const Json* pbr = mat ? mat->get("pbrMetallicRoughness") : nullptr;
        auto vecn = [](const Json* j, const char* key, float* dst, int n, float dflt) {
            const Json* a = j ? j->get(key) : nullptr;
            for (int i = 0; i < n; i++) dst[i] = (a && a->kind == Json::Arr && (size_t)i < a->arr.size() && a->arr[i].kind == Json::Num) ? (float)a->arr[i].num : dflt;
        };
        vecn(pbr, "baseColorFactor", r.m.base_color_value, 4, 1.0f);
        r.m.metallic_factor = pbr ? (float)pbr->number("metallicFactor", 1.0) : 1.0f;
        r.m.roughness_factor = pbr ? (float)pbr->number("roughnessFactor", 1.0) : 1.0f;
        vecn(mat, "emissiveFactor", r.m.emissive_factor, 3, 0.0f);
        const Json* ext = mat ? mat->get("extensions") : nullptr;
        const Json* es = ext ? ext->get("KHR_materials_emissive_strength") : nullptr;
        r.m.emissive_factor[3] = es ? (float)es->number("emissiveStrength", 1.0) : 0.0f;   // absent extension -> 0.0 (gltf/mod.rs:222)
        const Json* tr = ext ? ext->get("KHR_materials_transmission") : nullptr;
        r.m.transmission_factor = tr ? (float)tr->number("transmissionFactor", 0.0) : 0.0f;
        const Json* ior = ext ? ext->get("KHR_materials_ior") : nullptr;
        r.m.ior = ior ? (float)ior->number("ior", 1.5) : 1.5f;
        r.m.alpha_mode = 0; r.m.alpha_cutoff = 0.0f;                                        // resources/material.rs:74-75
        uint32_t* slots = &r.m.base_color_image;
        for (int i = 0; i < 10; i++) slots[i] = SR_NULL_TEXTURE;
        texture_ref(pbr, "baseColorTexture", r.m.base_color_image, r.sets[0]);
        texture_ref(pbr, "metallicRoughnessTexture", r.m.metallic_roughness_image, r.sets[1]);
        texture_ref(mat, "normalTexture", r.m.normal_image, r.sets[2]);
        texture_ref(mat, "occlusionTexture", r.m.occlusion_image, r.sets[3]);
        texture_ref(mat, "emissiveTexture", r.m.emissive_image, r.sets[4]);
        r.is_emissive = r.m.emissive_factor[3] > 0.0f || r.m.emissive_factor[0] != 0.0f || r.m.emissive_factor[1] != 0.0f || r.m.emissive_factor[2] != 0.0f;  // :272
        return r;
    }

    struct Primitive { PrimKey key; SrMaterial material; std::vector<SrEmissiveTriangle> local_emissive; };

    bool process_mesh(long mesh_index, std::vector<Primitive>& prims) {
        const Json* mesh = element("meshes", mesh_index);
        if (!mesh) return fail("node refers to a missing mesh");
        const Json* plist = mesh->get("primitives");
        long i = 0;    // enumerate() over the SUPPORTED primitives (gltf/mod.rs:199)
        for (size_t pi = 0; plist && pi < plist->size(); pi++) {
            const Json& prim = plist->arr[pi];
            if ((long)prim.number("mode", 4) != 4) continue;                         // is_primitive_supported (:362-372)
            const Json* attrs = prim.get("attributes");
            const long pos_acc = attrs ? attrs->index("POSITION") : -1;
            if (pos_acc < 0) return fail("primitive without POSITION");              // .unwrap() in the reference (:203)
            const long idx_acc = prim.index("indices");
            Primitive p;
            p.key = PrimKey(pos_acc, idx_acc >= 0 ? idx_acc : i);                    // :207-212
            MaterialInfo mi = material_of(prim);
            p.material = mi.m;
            std::vector<float> positions; size_t n_pos = 0;
            if (mi.is_emissive || !primitive_data_map.count(p.key)) { if (!read_floats(pos_acc, 3, positions, &n_pos)) return false; }
            if (mi.is_emissive) {                                                    // :274-296
                std::vector<uint32_t> idx;
                if (idx_acc >= 0) { if (!read_indices(idx_acc, idx)) return false; }
                else { idx.resize(n_pos); for (size_t k = 0; k < n_pos; k++) idx[k] = (uint32_t)k; }
                const float e[3] = {mi.m.emissive_factor[0] * mi.m.emissive_factor[3], mi.m.emissive_factor[1] * mi.m.emissive_factor[3],
                                    mi.m.emissive_factor[2] * mi.m.emissive_factor[3]};     // scene.rs:117-122
                for (size_t k = 0; k + 2 < idx.size(); k += 3) {
                    if (idx[k] >= n_pos || idx[k + 1] >= n_pos || idx[k + 2] >= n_pos) return fail("index out of range");
                    SrEmissiveTriangle t;
                    memset(&t, 0, sizeof(t));
                    memcpy(t.v0, &positions[3 * idx[k]], 12); memcpy(t.v1, &positions[3 * idx[k + 1]], 12); memcpy(t.v2, &positions[3 * idx[k + 2]], 12);
                    memcpy(t.emission, e, 12);
                    p.local_emissive.push_back(t);
                }
            }
            if (!primitive_data_map.count(p.key)) {                                  // :301-351
                PrimData d;
                std::vector<float> normals, tangents; size_t n_nrm = 0, n_tan = 0;
                const long nrm_acc = attrs->index("NORMAL");
                if (nrm_acc < 0) return fail("primitive without NORMAL (read_normals().unwrap(), gltf/mod.rs:307)");
                if (!read_floats(nrm_acc, 3, normals, &n_nrm)) return false;
                if (n_nrm < n_pos) return fail("NORMAL shorter than POSITION");
                const long tan_acc = attrs->index("TANGENT");
                if (tan_acc >= 0) { if (!read_floats(tan_acc, 4, tangents, &n_tan)) return false; if (n_tan < n_pos) return fail("TANGENT shorter than POSITION"); }
                if (n_pos == 0) return fail("primitive without vertices");
                d.vertices.resize(n_pos);
                memset(d.vertices.data(), 0, n_pos * sizeof(SrVertex));
                for (size_t k = 0; k < n_pos; k++) {
                    memcpy(d.vertices[k].position, &positions[3 * k], 12);
                    memcpy(d.vertices[k].normal, &normals[3 * k], 12);
                    if (tan_acc >= 0) memcpy(d.vertices[k].tangent, &tangents[4 * k], 16);
                }
                if (idx_acc >= 0) { if (!read_indices(idx_acc, d.indices)) return false; }
                else { d.indices.resize(n_pos / 3); for (size_t k = 0; k < d.indices.size(); k++) d.indices[k] = (uint32_t)k; }   // sic: 0..len/3 (:330)
                for (uint32_t ix : d.indices) if (ix >= n_pos) return fail("index out of range");
                for (int set = 0; set < 5; set++) {                                  // insert_tex_coords! (:338-342)
                    char name[24];
                    snprintf(name, sizeof(name), "TEXCOORD_%ld", mi.sets[set]);
                    const long uv_acc = attrs->index(name);
                    if (uv_acc < 0) return fail(std::string("primitive without ") + name + " (read_tex_coords().unwrap(), gltf/mod.rs:38-46)");
                    std::vector<float> uv; size_t n_uv = 0;
                    if (!read_floats(uv_acc, 2, uv, &n_uv)) return false;
                    for (size_t k = 0; k < n_pos && k < n_uv; k++) {
                        float* dst = set == 0 ? d.vertices[k].base_color_tex_coord : set == 1 ? d.vertices[k].metallic_roughness_tex_coord
                                   : set == 2 ? d.vertices[k].normal_tex_coord : set == 3 ? d.vertices[k].occlusion_tex_coord : d.vertices[k].emissive_tex_coord;
                        dst[0] = uv[2 * k]; dst[1] = uv[2 * k + 1];
                    }
                }
                primitive_data_map.emplace(p.key, std::move(d));
            }
            prims.push_back(std::move(p));
            i++;
        }
        return true;
    }

    static Mat4 node_matrix(const Json& node) {
        Mat4 r = identity();
        const Json* m = node.get("matrix");
        if (m && m->kind == Json::Arr && m->arr.size() == 16) {                      // column-major in the file
            for (int c = 0; c < 4; c++) for (int row = 0; row < 4; row++) r.m[row * 4 + c] = (float)m->arr[c * 4 + row].num;
            return r;
        }
        float t[3] = {0, 0, 0}, q[4] = {0, 0, 0, 1}, s[3] = {1, 1, 1};
        auto rd = [&](const char* key, float* dst, int n) {
            const Json* a = node.get(key);
            if (a && a->kind == Json::Arr && (int)a->arr.size() == n) for (int i = 0; i < n; i++) dst[i] = (float)a->arr[i].num;
        };
        rd("translation", t, 3); rd("rotation", q, 4); rd("scale", s, 3);
        // matrix = translation * rotation * scale (gltf/mod.rs:170-172). Quaternion (x,y,z,w) -> rotation as in the
        // gltf crate's math (cgmath's formula): x2 = x+x, ... ; fp32 throughout.
        const float x = q[0], y = q[1], z = q[2], w = q[3];
        const float x2 = x + x, y2 = y + y, z2 = z + z;
        const float xx2 = x2 * x, xy2 = x2 * y, xz2 = x2 * z, yy2 = y2 * y, yz2 = y2 * z, zz2 = z2 * z;
        const float sy2 = y2 * w, sz2 = z2 * w, sx2 = x2 * w;
        const float R[9] = {1.0f - yy2 - zz2, xy2 - sz2, xz2 + sy2,
                            xy2 + sz2, 1.0f - xx2 - zz2, yz2 - sx2,
                            xz2 - sy2, yz2 + sx2, 1.0f - xx2 - yy2};
        for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) r.m[i * 4 + j] = R[i * 3 + j] * s[j]; r.m[i * 4 + 3] = t[i]; }
        return r;
    }

    bool explore(long node_index, const Mat4& parent, int depth) {
        if (depth > 256) return fail("node hierarchy too deep (cycle?)");
        const Json* node = element("nodes", node_index);
        if (!node) return fail("scene refers to a missing node");
        if (node->has("camera")) return fail("nodes with cameras are not supported (todo!() in gltf/mod.rs:181-183)", SR_ERR_UNSUPPORTED);
        const Json* ext = node->get("extensions");
        if (ext && ext->has("KHR_lights_punctual")) return fail("nodes with lights are not supported (todo!() in gltf/mod.rs:185-187)", SR_ERR_UNSUPPORTED);
        const Mat4 transform = mul(parent, node_matrix(*node));
        const long mesh_index = node->index("mesh");
        if (mesh_index >= 0) {
            std::vector<Primitive> prims;
            if (!process_mesh(mesh_index, prims)) return false;
            for (Primitive& p : prims) {                                             // Scene::explore_node (scene.rs:103-167)
                uint32_t blas_index;
                auto it = primitives_blas_index.find(p.key);
                if (it != primitives_blas_index.end()) blas_index = it->second;
                else {
                    auto dit = primitive_data_map.find(p.key);
                    if (dit == primitive_data_map.end()) return fail("primitive data consumed twice");
                    GltfBlas b;
                    b.vertices = std::move(dit->second.vertices);
                    b.indices = std::move(dit->second.indices);
                    b.material = p.material;
                    b.emissive = p.local_emissive;
                    blas_index = (uint32_t)out.blases.size();
                    out.blases.push_back(std::move(b));
                    primitives_blas_index[p.key] = blas_index;
                }
                SrTransform t;
                memcpy(t.m, transform.m, 48);                                        // na_mat4_to_vk_transform (utils.rs:67-74)
                out.instances.emplace_back(blas_index, t);
            }
        }
        const Json* children = node->get("children");
        for (size_t c = 0; children && c < children->size(); c++)
            if (!explore(children->arr[c].kind == Json::Num && children->arr[c].num >= 0.0 && children->arr[c].num < 2147483648.0 ? (long)children->arr[c].num : -1, transform, depth + 1)) return false;
        return true;
    }

    bool build() {
        // samplers (gltf/mod.rs:90-99, scene.rs:68-83): absent filters default to LINEAR, wrap defaults to REPEAT (10497)
        const Json* smp = doc.get("samplers");
        for (size_t i = 0; smp && i < smp->size(); i++) {
            const Json& s = smp->arr[i];
            auto filt = [](long v, uint32_t dflt) -> uint32_t { return v < 0 ? dflt : ((v == 9728 || v == 9984 || v == 9986) ? SR_FILTER_NEAREST : SR_FILTER_LINEAR); };
            auto wrap = [](long v) -> uint32_t { return v == 33071 ? SR_ADDRESS_CLAMP_TO_EDGE : v == 33648 ? SR_ADDRESS_MIRRORED_REPEAT : SR_ADDRESS_REPEAT; };
            SrSamplerDesc d;
            d.min_filter = filt(s.index("minFilter"), SR_FILTER_LINEAR);
            d.mag_filter = filt(s.index("magFilter"), SR_FILTER_LINEAR);
            d.address_mode_u = wrap(s.index("wrapS"));
            d.address_mode_v = wrap(s.index("wrapT"));
            out.samplers.push_back(d);
        }
        const Json* tex = doc.get("textures");
        for (size_t i = 0; tex && i < tex->size(); i++) {
            const long src = tex->arr[i].index("source");
            if (src < 0) return fail("texture without source");
            out.textures.push_back(GltfTexture{(int32_t)tex->arr[i].index("sampler"), (uint32_t)src});
        }
        const Json* imgs = doc.get("images");
        for (size_t i = 0; imgs && i < imgs->size(); i++) {
            std::vector<uint8_t> bytes;
            const Json* uri = imgs->arr[i].get("uri");
            if (uri && uri->kind == Json::Str) { if (!resolve_uri(uri->str, bytes)) return false; }
            else {
                const Json* bv = element("bufferViews", imgs->arr[i].index("bufferView"));
                if (!bv) return fail("image without uri or bufferView");
                const long bi = bv->index("buffer");
                const size_t off = bv->size_value("byteOffset"), len = bv->size_value("byteLength");
                if (bi < 0 || (size_t)bi >= buffers.size() || off + len > buffers[bi].size()) return fail("image bufferView exceeds its buffer");
                bytes.assign(buffers[bi].begin() + off, buffers[bi].begin() + off + len);
            }
            DecodedImage im;
            std::string ierr;
            if (!srh::decode_image(bytes.data(), bytes.size(), im.w, im.h, im.channels, im.pixels, ierr)) return fail(ierr, SR_ERR_UNSUPPORTED);
            out.images.push_back(std::move(im));
        }
        for (const auto& t : out.textures) {
            if (t.source >= out.images.size()) return fail("texture refers to a missing image");
            if (t.sampler >= (int32_t)out.samplers.size()) return fail("texture refers to a missing sampler");
        }
        // default scene, or scene 0 (gltf/mod.rs:69-77)
        long scene_index = doc.index("scene");
        if (scene_index < 0) scene_index = 0;
        const Json* scene = element("scenes", scene_index);
        if (!scene) { char b[64]; snprintf(b, sizeof(b), "No scene with index: %ld found", scene_index); return fail(b); }
        const Json* roots = scene->get("nodes");
        for (size_t i = 0; roots && i < roots->size(); i++)
            if (!explore(roots->arr[i].kind == Json::Num && roots->arr[i].num >= 0.0 && roots->arr[i].num < 2147483648.0 ? (long)roots->arr[i].num : -1, identity(), 0)) return false;
        for (const auto& b : out.blases) {
            const uint32_t* slots = &b.material.base_color_image;
            for (int k = 0; k < 10; k += 2) if (slots[k] != SR_NULL_TEXTURE && slots[k] >= out.textures.size()) return fail("material refers to a missing texture");
        }
        return true;
    }
};

}  // namespace srh

struct SrGltf { srh::GltfScene scene; };

namespace srh {
bool decode_image(const uint8_t* data, size_t n, uint32_t& width, uint32_t& height, uint32_t& channels, std::vector<uint8_t>& pixels, std::string& err) {
    if (n >= 3 && data[0] == 0xFF && data[1] == 0xD8) return decode_jpeg(data, n, width, height, channels, pixels, err);   // gltf::import sniffs the content too
    DecodedImage im;
    if (!decode_png(data, n, im, err)) return false;
    width = im.w; height = im.h; channels = im.channels; pixels.swap(im.pixels);
    return true;
}
// image::load_from_memory(bytes).to_rgba8() (lib.rs:281-283, the embedded blue-noise PNG — 16-bit greyscale in the reference):
// any supported PNG / JPEG widened to RGBA8 — grey replicated to r, g, b; missing alpha = 255; 16-bit samples narrowed.
bool decode_image_rgba8(const uint8_t* data, size_t n, uint32_t& width, uint32_t& height, std::vector<uint8_t>& rgba, std::string& err) {
    uint32_t ch = 0;
    std::vector<uint8_t> px;
    if (n >= 3 && data[0] == 0xFF && data[1] == 0xD8) { if (!decode_jpeg(data, n, width, height, ch, px, err)) return false; }
    else {
        DecodedImage im;
        if (!decode_png(data, n, im, err, true)) return false;
        width = im.w; height = im.h; ch = im.channels; px.swap(im.pixels);
    }
    rgba.resize((size_t)width * height * 4);
    for (size_t i = 0; i < (size_t)width * height; i++) {
        const uint8_t* p = &px[i * ch];
        uint8_t* q = &rgba[i * 4];
        if (ch <= 2) { q[0] = q[1] = q[2] = p[0]; q[3] = ch == 2 ? p[1] : 255; }
        else { q[0] = p[0]; q[1] = p[1]; q[2] = p[2]; q[3] = ch == 4 ? p[3] : 255; }
    }
    return true;
}
}  // namespace srh

extern "C" {

// The image decoder of the glTF loader on its own (PNG 8-bit, baseline JPEG): extent and channel count, and — when `pixels`
// is non-null and `cap` is large enough — the w * h * channels bytes Image::new_from_data would be handed.
int sr_decode_image(const uint8_t* data, size_t n, uint32_t* width, uint32_t* height, uint32_t* channels, uint8_t* pixels, size_t cap) {
    if (!data || !width || !height || !channels) return srh::set_error(SR_ERR_INVALID_ARG, "sr_decode_image: null argument");
    std::vector<uint8_t> px;
    std::string err;
    if (!srh::decode_image(data, n, *width, *height, *channels, px, err)) return srh::set_error(SR_ERR_UNSUPPORTED, err);
    if (pixels) {
        if (cap < px.size()) return srh::set_error(SR_ERR_INVALID_ARG, "sr_decode_image: output too small");
        memcpy(pixels, px.data(), px.size());
    }
    return SR_OK;
}

// image::load_from_memory(..).to_rgba8(): what the reference does with its embedded blue-noise texture (lib.rs:281-284) and what
// a host does with any image it wants as RGBA8. 16-bit PNG samples are narrowed as image-rs does; w * h * 4 bytes.
int sr_decode_image_rgba8(const uint8_t* data, size_t n, uint32_t* width, uint32_t* height, uint8_t* pixels, size_t cap) {
    if (!data || !width || !height) return srh::set_error(SR_ERR_INVALID_ARG, "sr_decode_image_rgba8: null argument");
    std::vector<uint8_t> px;
    std::string err;
    if (!srh::decode_image_rgba8(data, n, *width, *height, px, err)) return srh::set_error(SR_ERR_UNSUPPORTED, err);
    if (pixels) {
        if (cap < px.size()) return srh::set_error(SR_ERR_INVALID_ARG, "sr_decode_image_rgba8: output too small");
        memcpy(pixels, px.data(), px.size());
    }
    return SR_OK;
}

// Gltf::new + create_default_scene + the CPU side of Scene::load_into_gpu
int sr_gltf_open(const char* path, SrGltf** out) {
    if (!path || !out) return srh::set_error(SR_ERR_INVALID_ARG, "sr_gltf_open: null argument");
    srh::GltfLoader L;
    if (!L.open(path) || !L.build()) return srh::set_error(L.err_code, L.err);
    SrGltf* g = new SrGltf();
    g->scene = std::move(L.out);
    *out = g;
    return SR_OK;
}

int sr_gltf_close(SrGltf* g) { delete g; return SR_OK; }

int sr_gltf_counts(const SrGltf* g, uint32_t* n_blases, uint32_t* n_instances, uint32_t* n_images, uint32_t* n_samplers, uint32_t* n_textures) {
    if (!g) return srh::set_error(SR_ERR_INVALID_ARG, "sr_gltf_counts: null argument");
    if (n_blases) *n_blases = (uint32_t)g->scene.blases.size();
    if (n_instances) *n_instances = (uint32_t)g->scene.instances.size();
    if (n_images) *n_images = (uint32_t)g->scene.images.size();
    if (n_samplers) *n_samplers = (uint32_t)g->scene.samplers.size();
    if (n_textures) *n_textures = (uint32_t)g->scene.textures.size();
    return SR_OK;
}

int sr_gltf_blas(const SrGltf* g, uint32_t i, const SrVertex** vertices, uint32_t* n_vertices, const uint32_t** indices, uint32_t* n_indices,
                 SrMaterial* material, const SrEmissiveTriangle** emissive, uint32_t* n_emissive) {
    if (!g || i >= g->scene.blases.size()) return srh::set_error(SR_ERR_INVALID_ARG, "sr_gltf_blas: index out of range");
    const srh::GltfBlas& b = g->scene.blases[i];
    if (vertices) *vertices = b.vertices.data();
    if (n_vertices) *n_vertices = (uint32_t)b.vertices.size();
    if (indices) *indices = b.indices.data();
    if (n_indices) *n_indices = (uint32_t)b.indices.size();
    if (material) *material = b.material;
    if (emissive) *emissive = b.emissive.data();
    if (n_emissive) *n_emissive = (uint32_t)b.emissive.size();
    return SR_OK;
}

int sr_gltf_instance(const SrGltf* g, uint32_t i, uint32_t* blas_index, SrTransform* transform) {
    if (!g || i >= g->scene.instances.size()) return srh::set_error(SR_ERR_INVALID_ARG, "sr_gltf_instance: index out of range");
    if (blas_index) *blas_index = g->scene.instances[i].first;
    if (transform) *transform = g->scene.instances[i].second;
    return SR_OK;
}

int sr_gltf_image(const SrGltf* g, uint32_t i, const uint8_t** pixels, uint32_t* width, uint32_t* height, uint32_t* channels) {
    if (!g || i >= g->scene.images.size()) return srh::set_error(SR_ERR_INVALID_ARG, "sr_gltf_image: index out of range");
    const DecodedImage& im = g->scene.images[i];
    if (pixels) *pixels = im.pixels.data();
    if (width) *width = im.w;
    if (height) *height = im.h;
    if (channels) *channels = im.channels;
    return SR_OK;
}

int sr_gltf_sampler(const SrGltf* g, uint32_t i, SrSamplerDesc* out) {
    if (!g || !out || i >= g->scene.samplers.size()) return srh::set_error(SR_ERR_INVALID_ARG, "sr_gltf_sampler: index out of range");
    *out = g->scene.samplers[i];
    return SR_OK;
}

int sr_gltf_texture(const SrGltf* g, uint32_t i, int32_t* sampler, uint32_t* source) {
    if (!g || i >= g->scene.textures.size()) return srh::set_error(SR_ERR_INVALID_ARG, "sr_gltf_texture: index out of range");
    if (sampler) *sampler = g->scene.textures[i].sampler;
    if (source) *source = g->scene.textures[i].source;
    return SR_OK;
}

}  // extern "C"
