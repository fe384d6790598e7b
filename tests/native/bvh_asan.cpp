// Sanitizer driver for the host BVH builder (tests/test_native_sanitizers.py): random and degenerate triangle sets through
// srh::build_bvh + tree_levels under -fsanitize=address,undefined.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>
#include "../../sunray_amd/csrc/host.h"
namespace srh { int set_error(int code, const std::string&) { return code; } }
int main() {
    unsigned rng = 12345;
    auto rnd = [&]() { rng ^= rng << 13; rng ^= rng >> 17; rng ^= rng << 5; return (float)(rng & 0xFFFFFF) / 16777216.0f; };
    const int sizes[] = {0, 1, 2, 3, 5, 64, 1000, 40000};
    for (int mode = 0; mode < 4; mode++)
        for (int n : sizes) {
            std::vector<srh::BuildTri> t(n);
            for (int i = 0; i < n; i++) {
                for (int a = 0; a < 3; a++) {
                    const float c = mode == 1 ? 0.5f : (mode == 2 ? (a == 0 ? (float)i * (float)i : 0.0f) : rnd() * 100.0f);   // all equal | a line | random
                    t[i].v0[a] = c; t[i].e1[a] = mode == 3 ? 0.0f : rnd() * 0.1f; t[i].e2[a] = mode == 3 ? 0.0f : rnd() * 0.1f;   // mode 3: points
                }
                t[i].prim = i; t[i].inst = 0; t[i].gid = i;
            }
            srh::BvhResult r;
            srh::build_bvh(t, 31, r);
            std::vector<uint32_t> ln, lo;
            srh::tree_levels(r.nodes, ln, lo);
            if (r.max_stack > 31 || r.tris.size() != (size_t)n * 12 || ln.size() != r.n_nodes) { printf("FAIL mode %d n %d\n", mode, n); return 1; }
        }
    printf("bvh ok\n");
    return 0;
}
