// lds_cost.hip -- what one LDS wave-instruction costs on a loaded CU (gfx950), for the access
// patterns of score_hist_kernel: pair/triple table lookups (ds_read_u16), row re-reads of the
// staged strip (ds_read_b32 / ds_read2_b32 at 19- and 20-byte pitch, W-dword pitch), the strip
// writes (ds_write_b128), histogram atomics (ds_add_u32 on random bins) and ds_bpermute.
// One 1024-thread workgroup per CU; every wave issues the same instruction REPS x 16 times with
// fixed per-lane addresses; cycles per wave-instruction per CU = s_memtime delta / (16 waves x n).
//   hipcc -O3 --offload-arch=gfx950 scripts/micro/lds_cost.hip -o scripts/micro/lds_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int kReps = 256;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define RD16(INSTR)                                                                                   \
    for (int it = 0; it < kReps; ++it) {                                                              \
        unsigned r0, r1, r2, r3, r4, r5, r6, r7;                                                      \
        asm volatile(INSTR " %0, %8\n\t" INSTR " %1, %9\n\t" INSTR " %2, %10\n\t" INSTR " %3, %11\n\t"    \
                     INSTR " %4, %8\n\t" INSTR " %5, %9\n\t" INSTR " %6, %10\n\t" INSTR " %7, %11\n\t"    \
                     "s_waitcnt lgkmcnt(0)"                                                           \
                     : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6),  \
                       "=&v"(r7)                                                                      \
                     : "v"(a0), "v"(a1), "v"(a2), "v"(a3)                                             \
                     : "memory");                                                                     \
        acc += r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;                                                 \
    }                                                                                                 \
    n_instr = 8 * kReps;

__global__ void __launch_bounds__(1024)
lds_cost_kernel(int pattern, const unsigned *__restrict__ rnd, unsigned long long *__restrict__ cycles,
                unsigned *__restrict__ sink)
{
    extern __shared__ unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 160 * 1024 / 4 - 64; i += 1024) reinterpret_cast<unsigned *>(smem)[i] = i * 2654435761u;
    __syncthreads();
    // four fixed per-lane random draws
    const unsigned q0 = rnd[(blockIdx.x * 1024 + tid) * 4 + 0], q1 = rnd[(blockIdx.x * 1024 + tid) * 4 + 1];
    const unsigned q2 = rnd[(blockIdx.x * 1024 + tid) * 4 + 2], q3 = rnd[(blockIdx.x * 1024 + tid) * 4 + 3];
    unsigned a0 = 0, a1 = 0, a2 = 0, a3 = 0, acc = 0;
    const unsigned strip = 2048u + (unsigned)wave * 6144u;   // a wave-private 6 KiB region
    long long n_instr = 0;
    auto pair_ix = [](unsigned q) { return 2u * ((q & 3u) + 8u * ((q >> 2) & 3u)); };          // 16 hot of 64 u16
    auto triple_ix = [](unsigned q) { return 2u * (q & 63u); };                                    // 64 u16 = 32 dwords
    auto quad_ix = [](unsigned q) { return 2u * (q & 255u); };                                     // 256 u16
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    switch (pattern) {
    case 0:  // pair-table lookup, four different tables (128 B apart)
        a0 = pair_ix(q0); a1 = 128 + pair_ix(q1); a2 = 256 + pair_ix(q2); a3 = 384 + pair_ix(q3);
        RD16("ds_read_u16") break;
    case 1:  // triple-table lookup (64 entries, one dword per bank)
        a0 = triple_ix(q0); a1 = 128 + triple_ix(q1); a2 = 256 + triple_ix(q2); a3 = 384 + triple_ix(q3);
        RD16("ds_read_u16") break;
    case 2:  // quad-table lookup (256 entries: 4 dwords per bank)
        a0 = quad_ix(q0); a1 = 512 + quad_ix(q1); a2 = 1024 + quad_ix(q2); a3 = 1536 + quad_ix(q3);
        RD16("ds_read_u16") break;
    case 3:  // row re-read, 19-byte pitch, aligned down (today's kernel), ds_read_b32
        a0 = strip + ((19u * lane) & ~3u); a1 = a0 + 4; a2 = a0 + 8; a3 = a0 + 12;
        RD16("ds_read_b32") break;
    case 4:  // row re-read, 20-byte pitch (odd dword pitch: conflict free)
        a0 = strip + 20u * lane; a1 = a0 + 4; a2 = a0 + 8; a3 = a0 + 12;
        RD16("ds_read_b32") break;
    case 5:  // four rows per lane: pitch 19 dwords
        a0 = strip + 76u * lane; a1 = a0 + 4; a2 = a0 + 8; a3 = a0 + 12;
        RD16("ds_read_b32") break;
    case 6:  // pair-table lookup with dword entries
        a0 = 2 * pair_ix(q0); a1 = 256 + 2 * pair_ix(q1); a2 = 512 + 2 * pair_ix(q2); a3 = 768 + 2 * pair_ix(q3);
        RD16("ds_read_b32") break;
    case 7:  // ds_read_b64, 8-byte linear
        a0 = strip + 8u * lane; a1 = a0 + 512; a2 = a0 + 1024; a3 = a0 + 1536;
        for (int it = 0; it < kReps; ++it) {
            unsigned long long r0, r1, r2, r3;
            asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %6\n\tds_read_b64 %3, %7\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
            acc += (unsigned)(r0 ^ r1 ^ r2 ^ r3);
        }
        n_instr = 4 * kReps;
        break;
    case 8:  // ds_read_b128 linear
        a0 = strip + 16u * lane; a1 = a0 + 1024; a2 = a0 + 2048; a3 = a0 + 3072;
        for (int it = 0; it < kReps; ++it) {
            u32x4 r0, r1, r2, r3;
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
            acc += r0.x ^ r1.y ^ r2.z ^ r3.w;
        }
        n_instr = 4 * kReps;
        break;
    case 9:  // ds_write_b128 linear (the strip staging)
        a0 = strip + 16u * lane; a1 = a0 + 1024; a2 = a0 + 2048; a3 = a0 + 3072;
        for (int it = 0; it < kReps; ++it) {
            u32x4 v = {q0 + it, q1, q2, q3};
            asm volatile("ds_write_b128 %0, %4\n\tds_write_b128 %1, %4\n\tds_write_b128 %2, %4\n\tds_write_b128 %3, %4\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(v) : "memory");
        }
        n_instr = 4 * kReps;
        break;
    case 10:  // ds_write_b32 linear
        a0 = strip + 4u * lane; a1 = a0 + 256; a2 = a0 + 512; a3 = a0 + 768;
        for (int it = 0; it < kReps; ++it) {
            unsigned v = q0 + it;
            asm volatile("ds_write_b32 %0, %4\n\tds_write_b32 %1, %4\n\tds_write_b32 %2, %4\n\tds_write_b32 %3, %4\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(v) : "memory");
        }
        n_instr = 4 * kReps;
        break;
    case 11:  // histogram: ds_add_u32 on random bins of a 7425-bin window (uniform -- worst case)
    case 12:  // histogram: bins drawn from a narrow bell (sum of 4 draws): hot bins collide more
    case 13: {  // ds_add_u32, all lanes distinct consecutive dwords (no conflict)
        const unsigned hb = 102400u;   // histogram base (above the strips)
        if (pattern == 11) {
            a0 = hb + 4u * (q0 % 7425u); a1 = hb + 4u * (q1 % 7425u); a2 = hb + 4u * (q2 % 7425u); a3 = hb + 4u * (q3 % 7425u);
        } else if (pattern == 12) {
            auto bell = [](unsigned q) { return ((q & 511u) + ((q >> 9) & 511u) + ((q >> 18) & 511u) + ((q >> 23) & 511u)) + 2000u; };
            a0 = hb + 4u * bell(q0); a1 = hb + 4u * bell(q1); a2 = hb + 4u * bell(q2); a3 = hb + 4u * bell(q3);
        } else {
            a0 = hb + 4u * lane; a1 = a0 + 256; a2 = a0 + 512; a3 = a0 + 768;
        }
        for (int it = 0; it < kReps; ++it) {
            unsigned one = 1u;
            asm volatile("ds_add_u32 %0, %4\n\tds_add_u32 %1, %4\n\tds_add_u32 %2, %4\n\tds_add_u32 %3, %4\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(one) : "memory");
        }
        n_instr = 4 * kReps;
        break;
    }
    case 14:  // ds_bpermute_b32 (crossbar only)
        a0 = 4u * ((lane * 5u) & 63u); a1 = 4u * ((lane * 7u + 3u) & 63u); a2 = 4u * (lane ^ 1u); a3 = 4u * (63u - lane);
        for (int it = 0; it < kReps; ++it) {
            unsigned r0, r1, r2, r3;
            asm volatile("ds_bpermute_b32 %0, %4, %8\n\tds_bpermute_b32 %1, %5, %8\n\tds_bpermute_b32 %2, %6, %8\n\t"
                         "ds_bpermute_b32 %3, %7, %8\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(q0) : "memory");
            acc += r0 ^ r1 ^ r2 ^ r3;
        }
        n_instr = 4 * kReps;
        break;
    case 15:  // ds_read2_b32 at 19-byte pitch (today's row read: 3 per row)
        a0 = strip + ((19u * lane) & ~3u); a1 = a0 + 8; a2 = a0 + 16; a3 = a0 + 2432;
        for (int it = 0; it < kReps; ++it) {
            unsigned long long r0, r1, r2, r3;
            asm volatile("ds_read2_b32 %0, %4 offset1:1\n\tds_read2_b32 %1, %5 offset1:1\n\tds_read2_b32 %2, %6 offset1:1\n\t"
                         "ds_read2_b32 %3, %7 offset1:1\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
            acc += (unsigned)(r0 ^ r1 ^ r2 ^ r3);
        }
        n_instr = 4 * kReps;
        break;
    case 16:  // ds_read2_b32, four rows per lane (pitch 19 dwords)
        a0 = strip + 76u * lane; a1 = a0 + 8; a2 = a0 + 16; a3 = a0 + 24;
        for (int it = 0; it < kReps; ++it) {
            unsigned long long r0, r1, r2, r3;
            asm volatile("ds_read2_b32 %0, %4 offset1:1\n\tds_read2_b32 %1, %5 offset1:1\n\tds_read2_b32 %2, %6 offset1:1\n\t"
                         "ds_read2_b32 %3, %7 offset1:1\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
            acc += (unsigned)(r0 ^ r1 ^ r2 ^ r3);
        }
        n_instr = 4 * kReps;
        break;
    case 17:  // ds_read_u8 pair lookup (byte tables)
        a0 = pair_ix(q0) / 2; a1 = 64 + pair_ix(q1) / 2; a2 = 128 + pair_ix(q2) / 2; a3 = 192 + pair_ix(q3) / 2;
        RD16("ds_read_u8") break;
    default: break;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    if (tid == 0) {
        cycles[blockIdx.x * 2 + 0] = t2 - t0;        // whole workgroup
        cycles[blockIdx.x * 2 + 1] = (unsigned long long)n_instr;
    }
    (void)t1;
    if (acc == 0x12345u) sink[tid] = acc;
}

int main()
{
    const int blocks = 256;
    std::vector<unsigned> h_rnd((size_t)blocks * 1024 * 4);
    unsigned long long s = 88172645463325252ull;
    for (auto &x : h_rnd) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (unsigned)(s >> 16); }
    unsigned *d_rnd, *d_sink;
    unsigned long long *d_cyc;
    (void)hipMalloc(&d_rnd, h_rnd.size() * 4);
    (void)hipMalloc(&d_sink, 4096);
    (void)hipMalloc(&d_cyc, blocks * 2 * 8);
    (void)hipMemcpy(d_rnd, h_rnd.data(), h_rnd.size() * 4, hipMemcpyHostToDevice);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(lds_cost_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const char *names[] = {"pair lookup ds_read_u16 (16 hot / 64)", "triple lookup ds_read_u16 (64 in 32 dwords)",
                           "quad lookup ds_read_u16 (256)", "row read b32 pitch 19 B aligned down", "row read b32 pitch 20 B",
                           "row read b32 pitch 76 B (4 rows/lane)", "pair lookup ds_read_b32 entries", "ds_read_b64 linear",
                           "ds_read_b128 linear", "ds_write_b128 linear", "ds_write_b32 linear", "ds_add_u32 uniform 7425 bins",
                           "ds_add_u32 bell ~1000 bins", "ds_add_u32 linear", "ds_bpermute_b32", "ds_read2_b32 pitch 19 B",
                           "ds_read2_b32 pitch 76 B", "pair lookup ds_read_u8"};
    for (int p = 0; p < 18; ++p) {
        std::vector<unsigned long long> h(blocks * 2);
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(lds_cost_kernel, dim3(blocks), dim3(1024), 160 * 1024 - 256, 0, p, d_rnd, d_cyc, d_sink);
            (void)hipDeviceSynchronize();
        }
        (void)hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double sum = 0;
        for (int b = 0; b < blocks; ++b) sum += (double)h[b * 2] / ((double)h[b * 2 + 1] * 16.0);
        // s_memtime ticks at 100 MHz on gfx9xx? (guide: tick = shader cycle on gfx950) -- print raw
        printf("%-46s %7.2f ticks per wave-instruction per CU (n=%llu)\n", names[p], sum / blocks, h[1]);
    }
    return 0;
}
