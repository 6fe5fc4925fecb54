// Host-side sanitizer driver (CPU container, no GPU): links libyolop_asan.so (`make -C yolo-puncture_amd/csrc asan`) and walks the
// part of the C-ABI that runs without a device for every variant x task x dtype: graph build, weight hand-over (host copies),
// planning for several input shapes, op / tensor introspection, kernel-symbol completion, tune-cache round trip, per-shape tuning
// memo and the lane schedule (yp_debug_host_selftest), error paths, destroy. AddressSanitizer + UBSan abort on the first finding.
#include "../include/yolop.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x)                                                                                         \
    do {                                                                                                 \
        if (!(x)) { fprintf(stderr, "asan_host: CHECK failed: %s (%s:%d): %s\n", #x, __FILE__, __LINE__, yp_last_error()); exit(1); } \
    } while (0)

int main() {
    const char variants[] = {'n', 's', 'm', 'b', 'l', 'x'};
    const int shapes[][3] = {{1, 64, 64}, {2, 96, 128}, {1, 384, 640}, {32, 640, 640}, {8, 640, 640}, {3, 480, 608}, {1, 640, 480}, {5, 32, 32}};
    long launches = 0;
    const int families[] = {YP_FAMILY_V10, YP_FAMILY_V8, YP_FAMILY_11};
    for (int fam : families)
    for (char v : variants)
        for (int task = 0; task < 2; ++task)
            for (int dtype = 0; dtype < 2; ++dtype) {
                if (fam != YP_FAMILY_V10 && (task == 0 || v == 'b' || dtype == 1)) continue;     // the v8 / 11 families: n s m l x, segment
                yp_model_desc d{v, 80, task, dtype, 300, fam};
                yp_engine* e = nullptr;
                CHECK(yp_create(&d, 0, &e) == YP_OK);
                const int nw = yp_weight_count(e);
                CHECK(nw > 0);
                for (int i = 0; i < nw; ++i) {
                    char name[256];
                    int64_t shp[4];
                    int nd = 0;
                    CHECK(yp_weight_info(e, i, name, sizeof(name), shp, &nd) == YP_OK);
                    size_t n = 1;
                    for (int k = 0; k < nd; ++k) n *= (size_t)shp[k];
                    std::vector<float> w(n);
                    for (size_t k = 0; k < n; ++k) w[k] = (float)((k * 2654435761u) & 0xffff) / 65536.f - 0.5f;
                    CHECK(yp_set_weight(e, name, w.data(), shp, nd) == YP_OK);
                    if (i == 0) {                                       // error paths: wrong rank, wrong shape, unknown name
                        int64_t bad[4] = {shp[0] + 1, shp[1], shp[2], shp[3]};
                        CHECK(yp_set_weight(e, name, w.data(), bad, nd) < 0);
                        CHECK(yp_set_weight(e, name, w.data(), shp, 1) < 0);
                        CHECK(yp_set_weight(e, "model.99.weight", w.data(), shp, nd) < 0);
                    }
                }
                CHECK(yp_finalize(e) < 0);                              // no device here: must refuse, not fall back
                CHECK(yp_plan(e, 1, 100, 100) < 0);
                for (const auto& s : shapes) {
                    const int nops = yp_plan(e, s[0], s[1], s[2]);
                    CHECK(nops > 0);
                    for (int i = 0; i < nops; ++i) {
                        char name[8];                                   // deliberately short: snprintf must truncate
                        char kname[256];
                        int kind, t, co, c;
                        double fl, by;
                        CHECK(yp_op_info(e, i, name, sizeof(name), &kind, &fl, &by) == YP_OK);
                        CHECK(yp_op_kernel(e, i, kname, sizeof(kname)) == YP_OK);
                        CHECK(yp_op_output(e, i, &t, &co, &c) == YP_OK);
                    }
                    const int nt = yp_tensor_count(e);
                    for (int i = 0; i < nt; ++i) {
                        char name[256];
                        int dims[4], f32;
                        CHECK(yp_tensor_info(e, i, name, sizeof(name), dims, &f32) == YP_OK);
                    }
                    const int st = yp_debug_host_selftest(e);
                    CHECK(st > 0);
                    launches += st;
                }
                // back to the first shape: the per-shape memo path
                CHECK(yp_plan(e, shapes[0][0], shapes[0][1], shapes[0][2]) > 0);
                CHECK(yp_debug_host_selftest(e) > 0);
                CHECK(yp_forward(e, nullptr, 1, 64, 64, nullptr, nullptr, nullptr, nullptr) < 0);
                CHECK(yp_destroy(e) == YP_OK);
            }
    yp_model_desc bad{'q', 80, 0, 0, 300, 0};
    yp_engine* e = nullptr;
    CHECK(yp_create(&bad, 0, &e) < 0);
    printf("asan_host: ok (%ld scheduled launches walked)\n", launches);
    return 0;
}
