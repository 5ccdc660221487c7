// Entry points declared in include/ganq_hip.h whose kernels are not written yet.  They fail loudly.
#include "common.h"
using namespace ganq;

extern "C" int ganq_hessian_accum(float*, const void*, int, int64_t, int64_t, int64_t, int64_t, void*) {
    return fail(-9, "ganq_hessian_accum: not implemented yet");
}
extern "C" size_t ganq_kmeans_workspace_bytes(int64_t, int64_t, int) { return 0; }
extern "C" int ganq_kmeans_init(const float*, const double*, int64_t, int64_t, int, float*, void*, size_t, void*) {
    return fail(-9, "ganq_kmeans_init: not implemented yet");
}
extern "C" int ganq_lut_linear_fwd(const void*, const int32_t*, const void*, const void*, int, int64_t, int64_t, int64_t,
                                   int, void*, void*) {
    return fail(-9, "ganq_lut_linear_fwd: not implemented yet");
}
extern "C" int ganq_pack_indices(const uint8_t*, int64_t, int64_t, int, int32_t*, void*) {
    return fail(-9, "ganq_pack_indices: not implemented yet");
}
extern "C" int ganq_unpack_indices(const int32_t*, int64_t, int64_t, int, uint8_t*, void*) {
    return fail(-9, "ganq_unpack_indices: not implemented yet");
}
