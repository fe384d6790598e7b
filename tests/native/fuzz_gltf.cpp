// Sanitizer fuzz driver for the glTF loader (tests/test_native_sanitizers.py): mutates a seed file (byte flips, 32-bit
// overwrites, truncation) and parses every mutant. Built with g++ -fsanitize=address,undefined; usage: fuzz SEED ITERS RNG TMP.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/sunray_hip.h"
namespace srh { int set_error(int code, const std::string&) { return code; } }
int main(int argc, char** argv) {
    std::vector<unsigned char> seed;
    FILE* f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); seed.resize(n); fread(seed.data(), 1, n, f); fclose(f);
    unsigned iters = atoi(argv[2]), rng = atoi(argv[3]);
    auto next = [&]() { rng ^= rng << 13; rng ^= rng >> 17; rng ^= rng << 5; return rng; };
    unsigned ok = 0;
    for (unsigned it = 0; it < iters; it++) {
        std::vector<unsigned char> m = seed;
        unsigned k = it == 0 ? 0 : 1 + next() % 8;
        for (unsigned j = 0; j < k; j++) {
            unsigned pos = next() % m.size();
            switch (next() % 4) {
                case 0: m[pos] = (unsigned char)next(); break;
                case 1: m[pos] ^= 1u << (next() % 8); break;
                case 2: if (pos + 4 <= m.size()) { unsigned v = next() % 3 == 0 ? 0xFFFFFFFFu : next(); memcpy(&m[pos], &v, 4); } break;
                case 3: if (m.size() > 64) m.resize(m.size() - next() % 32); break;
            }
        }
        FILE* o = fopen(argv[4], "wb"); fwrite(m.data(), 1, m.size(), o); fclose(o);
        SrGltf* g = nullptr;
        if (sr_gltf_open(argv[4], &g) == 0) {
            ok++;
            uint32_t nb, ni, nim, ns, nt; sr_gltf_counts(g, &nb, &ni, &nim, &ns, &nt);
            for (uint32_t b = 0; b < nb; b++) { const SrVertex* v; const uint32_t* idx; uint32_t nv, nx, ne; const SrEmissiveTriangle* et; SrMaterial mat;
                sr_gltf_blas(g, b, &v, &nv, &idx, &nx, &mat, &et, &ne); volatile float s = 0; for (uint32_t q = 0; q < nx; q++) s += v[idx[q]].position[0]; }
            sr_gltf_close(g);
        }
    }
    printf("%s: %u iterations, %u parsed\n", argv[1], iters, ok);
    return 0;
}
