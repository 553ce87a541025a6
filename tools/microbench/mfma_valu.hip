// Do the matrix cores and the vector ALU of a SIMD run side by side on gfx950?  (development aid)
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu mfma_valu.hip
//
// Part 1, synthetic streams over one grid (8 waves per SIMD): FMA only, MFMA only (fp32
// 16x16x4), both interleaved in every wave, and two ways of giving each wave one role:
//   split by wave       odd waves of a workgroup MFMA, even waves FMA.  The four waves of a
//                       workgroup go to the four SIMDs of a CU, so this puts the two roles on
//                       DIFFERENT SIMDs: it does not test co-issue (kept to show that).
//   split by workgroup  odd workgroups MFMA, even workgroups FMA: every SIMD holds waves of both
//                       roles (checked with the hardware id register: SIMDs that saw both).
// If the pipes overlap, a split run takes about max(fma/2, mfma/2), not (fma + mfma)/2.
//
// Part 2, the primary sweep's arithmetic in the two forms a wave-specialised kernel would use:
//   VALU waves   the hot loop's own mix per lane and triangle (64 pixels: 18 FMA, 112 additions,
//                64 v_and_or, 1 compare).
//   MFMA waves   U and W' of 16 pixels x 16 triangles per v_mfma_f32_16x16x4_f32 ((X, Y, 1, 0) x
//                (p.x, p.y, p.z, 0)), then one v_and_or per pair on the outputs.
// Each alone on half the workgroups, then together; pairs/s of each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef float f4 __attribute__((ext_vector_type(4)));

// MODE0: 1 = FMA, 2 = MFMA, 3 = both in every wave, 4 = role by wave, 5 = role by workgroup
template <int MODE0>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b, unsigned* seen, int shift)
{
    const int MODE = MODE0 == 4 ? ((threadIdx.x >> 6) & 1 ? 2 : 1)
                                : (MODE0 == 5 ? ((blockIdx.x >> shift) & 1 ? 2 : 1) : MODE0);
    if (seen && (threadIdx.x & 63) == 0) {
        // HW_ID (register 4): wave 3:0, SIMD 5:4, pipe 7:6, CU 11:8, SH 12, SE 15:13 ... XCC 19:16 on gfx94x/95x
        const unsigned id = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
        const unsigned simd = (id >> 4) & 3, cu = (id >> 8) & 15, se = (id >> 13) & 7;
        unsigned xcc = 0;
        xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));     // XCC_ID register
        const unsigned slot = (((xcc & 15) * 8 + se) * 16 + cu) * 4 + simd;
        atomicOr(&seen[slot & 4095], (unsigned)MODE);
    }
    float v[8];
    f4 m[4];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
    for (int i = 0; i < 4; ++i) m[i] = (f4){a, b, a, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep) {
            if (MODE & 2) {
#pragma unroll
                for (int i = 0; i < 4; ++i) m[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, m[i], 0, 0, 0);
            }
            if (MODE & 1) {
#pragma unroll
                for (int r2 = 0; r2 < 4; ++r2)
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = __builtin_fmaf(v[i], a, b);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += m[i].x + m[i].y + m[i].z + m[i].w;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// Part 2.  roles: bit 0 = even workgroups run the VALU form, bit 1 = odd workgroups run the MFMA
// form; a workgroup whose role is off returns at once.
__global__ void __launch_bounds__(256) k_sweep_forms(float* out, const float4* tris, int ntris, int roles,
                                                     int mfma_reps, int shift)
{
    const bool mfma_role = (blockIdx.x >> shift) & 1;
    if (!((roles >> (mfma_role ? 1 : 0)) & 1)) return;
    __shared__ float4 tile[2][256];
    for (int i = threadIdx.x; i < 512; i += 256) tile[i >> 8][i & 255] = tris[i % ntris];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    int acc_all = 0;
    if (!mfma_role) {
        const float Y = (float)(lane >> 2), Xs0 = (float)(lane & 3) * 0.25f;
        for (int t = 0; t < ntris; ++t) {
            const float4 P = tile[0][t & 255], Q = tile[1][t & 255];
            const float yu = __builtin_fmaf(Y, P.y, P.z), yw = __builtin_fmaf(Y, Q.y, Q.z);
            float un = __builtin_fmaf(Xs0, P.x, yu), wn = __builtin_fmaf(Xs0, Q.x, yw);
            int acc = __float_as_int(un) & __float_as_int(wn);
#pragma unroll
            for (int r = 1; r < 64; ++r) {
                if (r % 8 == 0) {
                    un = __builtin_fmaf(Xs0 + (float)r, P.x, yu);
                    wn = __builtin_fmaf(Xs0 + (float)r, Q.x, yw);
                } else {
                    un += P.x;
                    wn += Q.x;
                }
                acc = (__float_as_int(un) & __float_as_int(wn)) | acc;
            }
            if (acc < 0) acc_all += 1;
        }
    } else {
        // A operand: (X, Y, 1, 0) of this lane's pixel (row lane % 16 of the 16x4 A block, k = lane / 16);
        // B operand: component k of triangle column lane % 16.  One MFMA = U (or W') of 16 pixels x 16
        // triangles; output register j of a lane = pixel row 4 * (lane / 16) + j, triangle lane % 16.
        const int k = lane >> 4;
        float Aop[4];       // four pixel blocks per lane: 64 pixels x 16 triangles per round
        for (int b = 0; b < 4; ++b)
            Aop[b] = k == 0 ? (float)(16 * b + (lane & 15)) : (k == 1 ? (float)(lane & 15) : (k == 2 ? 1.0f : 0.0f));
        const f4 zero = (f4){0, 0, 0, 0};
        // mfma_reps blocks of 64 pixels, so that a wave of either role runs about as long
        for (int rep = 0; rep < mfma_reps; ++rep)
        for (int t = 0; t < ntris; t += 16) {
            const float4 P = tile[0][(t + (lane & 15)) & 255], Q = tile[1][(t + (lane & 15)) & 255];
            const float Bp = k == 0 ? P.x : (k == 1 ? P.y : (k == 2 ? P.z : 0.0f));
            const float Bq = k == 0 ? Q.x : (k == 1 ? Q.y : (k == 2 ? Q.z : 0.0f));
            int acc = 0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const f4 U = __builtin_amdgcn_mfma_f32_16x16x4f32(Aop[b], Bp, zero, 0, 0, 0);
                const f4 W = __builtin_amdgcn_mfma_f32_16x16x4f32(Aop[b], Bq, zero, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc = (__float_as_int(U[j]) & __float_as_int(W[j])) | acc;
            }
            if (acc < 0) acc_all += 1;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = (float)acc_all;
}

template <int MODE0>
float run(float* out, int wgs, int iters, unsigned* seen = nullptr, int shift = 0)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE0>, dim3(wgs), dim3(256), 0, 0, out, iters / 4, 1.0001f, 0.5f, (unsigned*)nullptr, shift);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE0>, dim3(wgs), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f, seen, shift);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

constexpr int kMfmaReps = 24;
float run_forms(float* out, const float4* tris, int wgs, int ntris, int roles, int shift)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_sweep_forms, dim3(wgs), dim3(256), 0, 0, out, tris, ntris / 8, roles, kMfmaReps, shift);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k_sweep_forms, dim3(wgs), dim3(256), 0, 0, out, tris, ntris, roles, kMfmaReps, shift);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    float* out;
    unsigned* seen;
    const int wgs = 256 * 8, iters = 4000;
    (void)hipMalloc(&out, sizeof(float) * 256 * wgs);
    (void)hipMalloc(&seen, sizeof(unsigned) * 4096);
    const float f = run<1>(out, wgs, iters), m = run<2>(out, wgs, iters), both = run<3>(out, wgs, iters);
    // per iteration and wave: 128 v_fma (64 lanes x 2 flop), 16 MFMA (16x16x4: 2048 flop each)
    const double waves = wgs * 4.0;
    printf("part 1: synthetic streams, %d workgroups of 4 waves\n", wgs);
    printf("fma only            %8.3f ms  %.1f TFLOP/s\n", f, waves * iters * 128 * 128 / (f * 1e-3) / 1e12);
    printf("mfma only           %8.3f ms  %.1f TFLOP/s\n", m, waves * iters * 16 * 2048.0 / (m * 1e-3) / 1e12);
    printf("both in every wave  %8.3f ms  (sum %.3f, max %.3f)\n", both, f + m, f > m ? f : m);
    unsigned host[4096];
    const int shifts[] = {0, 2, 8, 9, 10};
    for (int mode = 4; mode <= 9; ++mode) {
        (void)hipMemset(seen, 0, sizeof host);
        const int shift = mode == 4 ? 0 : shifts[mode - 5];     // role of a workgroup = bit `shift` of its number
        const float split = mode == 4 ? run<4>(out, wgs, iters, seen) : run<5>(out, wgs, iters, seen, shift);
        (void)hipMemcpy(host, seen, sizeof host, hipMemcpyDeviceToHost);
        int simds = 0, mixed = 0;
        for (unsigned h : host) { simds += h != 0; mixed += h == 3; }
        char label[64];
        if (mode == 4) snprintf(label, sizeof label, "wave");
        else snprintf(label, sizeof label, "workgroup bit %d", shift);
        printf("split by %-16s %8.3f ms  half the waves each: serial (sum/2) %.3f, overlapped (max/2) %.3f; "
               "%d of %d SIMDs seen held both roles\n", label, split, (f + m) / 2,
               (f > m ? f : m) / 2, mixed, simds);
    }

    // part 2
    const int ntris = 1 << 16;
    float4 ht[512];
    for (int i = 0; i < 512; ++i) ht[i] = make_float4(0.001f * (i % 7 + 1), 0.002f * (i % 5 + 1), 1.0f + i, 0.f);
    float4* tris;
    (void)hipMalloc(&tris, sizeof ht);
    (void)hipMemcpy(tris, ht, sizeof ht, hipMemcpyHostToDevice);
    // per workgroup with a role: 4 waves x 64 lanes x 64 pixels x ntris pairs (VALU form);
    // 4 waves x 64 pixels x ntris pairs x kMfmaReps (MFMA form: a wave covers 64 pixels x 16 triangles per round)
    const double pv = (wgs / 2) * 4.0 * 64 * 64 * ntris, pm = (wgs / 2) * 4.0 * 64 * ntris * kMfmaReps;
    printf("part 2: primary-sweep forms, %d workgroups of each role, %d triangles\n", wgs / 2, ntris);
    for (int shift : shifts) {
        const float tv = run_forms(out, tris, wgs, ntris, 1, shift), tm = run_forms(out, tris, wgs, ntris, 2, shift),
                    tb = run_forms(out, tris, wgs, ntris, 3, shift);
        printf("role = workgroup bit %d:  VALU waves alone %8.3f ms %.3e pairs/s   MFMA waves alone %8.3f ms %.3e pairs/s   "
               "both %8.3f ms %.3e pairs/s  (serial %.3f ms, overlapped %.3f ms)\n", shift, tv, pv / (tv * 1e-3), tm,
               pm / (tm * 1e-3), tb, (pv + pm) / (tb * 1e-3), tv + tm, tv > tm ? tv : tm);
    }
    // the VALU form on every workgroup, for scale
    {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_sweep_forms, dim3(wgs), dim3(256), 0, 0, out, tris, ntris, 1, kMfmaReps, 30);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("VALU form on all %d workgroups: %8.3f ms %.3e pairs/s\n", wgs, ms, 2 * pv / (ms * 1e-3));
    }
    (void)hipFree(out);
    (void)hipFree(seen);
    (void)hipFree(tris);
    return 0;
}
