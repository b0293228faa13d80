// Probe (GPU box): semantics of `buffer_load_dwordx4 ... offen lds` on gfx950 that the conv kernels rely on:
//  (1) lane l of a wave writes LDS bytes [base + 16*l, +16)  (wave-uniform base in M0, lane-linear destination)
//  (2) a lane whose byte offset is out of the descriptor's range writes ZEROS (not "skipped"): this is what
//      gives zero padding of out-of-image taps for free.
// build: hipcc --offload-arch=gfx950 -O2 lds_dma_oob.hip -o /tmp/lds_dma_oob && /tmp/lds_dma_oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const float* p, float* o, int nbytes) {
    __shared__ __attribute__((aligned(16))) float sm[4 * 256];
    for (int i = threadIdx.x; i < 1024; i += 256) sm[i] = 7.f;   // sentinel
    __syncthreads();
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, nbytes, 0x00020000);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    unsigned voff = (unsigned)(wave * 1024 + (63 - lane) * 16);    // reversed source order: dest must stay lane-linear
    if (lane % 3 == 1) voff = 0xFFFFFFFFu;                          // out of range
    if (lane % 3 == 2) voff = (unsigned)nbytes - 8;                 // straddles the end of the buffer
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(sm + wave * 256), 16, (int)voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += 256) o[i] = sm[i];
}
int main() {
    const int n = 1024;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = 100.f + i;
    float *d, *o;
    hipMalloc(&d, n * 4); hipMalloc(&o, n * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    probe<<<1, 256>>>(d, o, n * 4);
    std::vector<float> r(n);
    hipMemcpy(r.data(), o, n * 4, hipMemcpyDeviceToHost);
    int bad_lin = 0, bad_oob = 0, bad_straddle = 0;
    for (int w = 0; w < 4; ++w)
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 4; ++e) {
                float got = r[w * 256 + l * 4 + e];
                if (l % 3 == 0) { float want = 100.f + w * 256 + (63 - l) * 4 + e; if (got != want) ++bad_lin; }
                else if (l % 3 == 1) { if (got != 0.f) ++bad_oob; }
                else { if (got != 0.f) ++bad_straddle; }
            }
    printf("lane-linear dest mismatches %d | OOB lanes not zero %d | straddling lanes not zero %d  (sample oob %g straddle %g %g %g %g)\n",
           bad_lin, bad_oob, bad_straddle, r[4], r[8], r[9], r[10], r[11]);
    return (bad_lin || bad_oob) ? 1 : 0;
}
