// Micro-test: does the bounds check of a raw buffer load on gfx950 include the SGPR offset?
// desc = (base, num_records = NREC bytes); load 8 bytes at soffset = SOFF, voffset = lane * 8.
// Prints the first lane that reads zeros.  If soffset is part of the check the answer is (NREC - SOFF) / 8,
// if it is not, NREC / 8.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
__global__ void k(const double *base, uint32_t nrec, uint32_t soff, double *out) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(base), 0, nrec, 0x00027000);
    const u32x2_t v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, threadIdx.x * 8u, soff, 0);
    out[threadIdx.x] = __hiloint2double((int)v.y, (int)v.x);
}
int main() {
    double *d, *o, h[256], r[64];
    for (int i = 0; i < 256; i++) h[i] = 1000.0 + i;
    hipMalloc(&d, sizeof h); hipMalloc(&o, sizeof r);
    hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    const uint32_t cases[][2] = {{256, 0}, {256, 128}, {512, 128}, {512, 384}, {640, 512}};
    for (auto &c : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, c[0], c[1], o);
        hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost);
        int first0 = 64;
        for (int i = 0; i < 64; i++) if (r[i] == 0.0) { first0 = i; break; }
        printf("nrec %u soff %u: lane0 reads %.0f, first zero lane %d (incl: %d, excl: %d)\n", c[0], c[1], r[0], first0,
               (int)(c[0] > c[1] ? (c[0] - c[1]) / 8 : 0), (int)(c[0] / 8 > 64 ? 64 : c[0] / 8));
    }
    return 0;
}
