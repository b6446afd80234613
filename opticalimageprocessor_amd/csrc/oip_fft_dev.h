// oip_fft_dev.h -- device-side building blocks of the FFT passes (complex helpers, radix
// butterflies, compile-time Stockham stages), shared by fft.hip and the fused cross-power +
// inverse row pass in phasecorr.hip.
#pragma once

#include <hip/hip_runtime.h>

namespace oipfft {

// The transform is not part of the bit-exact contract (its factorisation already differs from
// OpenCV's), so multiply-adds may fuse here -- and only here: the library is built with
// -ffp-contract=off and every other kernel rounds after each multiply and add.
#define OIP_FFT_FMA _Pragma("clang fp contract(fast)")

// Complex helpers on packed-f32 instructions.  A float2 lives in a 64-bit register pair, and the
// VOP3P modifiers pick (op_sel / op_sel_hi) and negate (neg_lo / neg_hi) the halves each result lane
// reads, so a complex product is two instructions and "add i times" is one -- the compiler gets the
// arithmetic right but builds the swapped operands with v_mov (a quarter of the VALU work of a stage).
typedef float oip_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ oip_v2f to_v(float2 a) { oip_v2f r = {a.x, a.y}; return r; }
__device__ __forceinline__ float2 to_f2(oip_v2f a) { return make_float2(a.x, a.y); }

#ifndef OIP_FFT_NO_ASM
__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    oip_v2f t, r;
    // t = (a.x b.x, a.x b.y);  r = (-a.y b.y + t.x, a.y b.x + t.y)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(to_v(a)), "v"(to_v(b)));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(to_v(a)), "v"(to_v(b)), "v"(t));
    return to_f2(r);
}
// m + (-i) u = (m.x + u.y, m.y - u.x)
__device__ __forceinline__ float2 cadd_rot(float2 m, float2 u)
{
    oip_v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(to_v(m)), "v"(to_v(u)));
    return to_f2(r);
}
// m - (-i) u = (m.x - u.y, m.y + u.x)
__device__ __forceinline__ float2 csub_rot(float2 m, float2 u)
{
    oip_v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(to_v(m)), "v"(to_v(u)));
    return to_f2(r);
}
#else
__device__ __forceinline__ float2 cmul(float2 a, float2 b)
{
    OIP_FFT_FMA
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd_rot(float2 m, float2 u) { return make_float2(m.x + u.y, m.y - u.x); }
__device__ __forceinline__ float2 csub_rot(float2 m, float2 u) { return make_float2(m.x - u.y, m.y + u.x); }
#endif
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmuli_neg(float2 a) { return make_float2(a.y, -a.x); }   // a * (-i)
__device__ __forceinline__ float2 cscale(float2 a, float s) { return make_float2(a.x * s, a.y * s); }

// forward butterflies, w = exp(-2 pi i / r)
__device__ __forceinline__ void bf2(float2 *x)
{
    float2 a = x[0], b = x[1];
    x[0] = cadd(a, b);
    x[1] = csub(a, b);
}
__device__ __forceinline__ void bf3(float2 *x)
{
    OIP_FFT_FMA
    const float s = 0.86602540378443864676f;
    float2 t = cadd(x[1], x[2]);
    float2 d = cscale(csub(x[1], x[2]), s);
    float2 m = make_float2(x[0].x - 0.5f * t.x, x[0].y - 0.5f * t.y);
    x[0] = cadd(x[0], t);
    x[1] = cadd_rot(m, d);
    x[2] = csub_rot(m, d);
}
__device__ __forceinline__ void bf4(float2 *x)
{
    float2 s02 = cadd(x[0], x[2]), d02 = csub(x[0], x[2]);
    float2 s13 = cadd(x[1], x[3]), d13 = csub(x[1], x[3]);
    x[0] = cadd(s02, s13);
    x[2] = csub(s02, s13);
    x[1] = cadd_rot(d02, d13);
    x[3] = csub_rot(d02, d13);
}
__device__ __forceinline__ void bf5(float2 *x)
{
    OIP_FFT_FMA
    const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
    const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
    float2 t1 = cadd(x[1], x[4]), t2 = cadd(x[2], x[3]);
    float2 t3 = csub(x[1], x[4]), t4 = csub(x[2], x[3]);
    float2 m1 = make_float2(x[0].x + c1 * t1.x + c2 * t2.x, x[0].y + c1 * t1.y + c2 * t2.y);
    float2 m2 = make_float2(x[0].x + c2 * t1.x + c1 * t2.x, x[0].y + c2 * t1.y + c1 * t2.y);
    float2 u1 = make_float2(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y);
    float2 u2 = make_float2(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y);
    x[0] = cadd(x[0], cadd(t1, t2));
    x[1] = cadd_rot(m1, u1);
    x[4] = csub_rot(m1, u1);
    x[2] = cadd_rot(m2, u2);
    x[3] = csub_rot(m2, u2);
}
__device__ __forceinline__ void bf8(float2 *x)
{
    OIP_FFT_FMA
    const float h = 0.70710678118654752440f;
    float2 e[4] = {x[0], x[2], x[4], x[6]};
    float2 o[4] = {x[1], x[3], x[5], x[7]};
    bf4(e);
    bf4(o);
    const float2 o1 = cscale(cadd_rot(o[1], o[1]), h);       // o1 * w8   = h (o.x + o.y, o.y - o.x)
    const float2 o3 = cscale(csub_rot(o[3], o[3]), h);       // -o3 * w8^3 = h (o.x - o.y, o.x + o.y)
    x[0] = cadd(e[0], o[0]);
    x[4] = csub(e[0], o[0]);
    x[1] = cadd(e[1], o1);
    x[5] = csub(e[1], o1);
    x[2] = cadd_rot(e[2], o[2]);                             // o2 * w8^2 = -i o2
    x[6] = csub_rot(e[2], o[2]);
    x[3] = csub(e[3], o3);
    x[7] = cadd(e[3], o3);
}
// ---- composite butterflies: R = A * B points in registers -----------------------------------------------------
// x[n], n = n1 + A n2  ->  X[B k1 + k2]:  B-point transforms over n2 for each n1, the twiddles w_R^(n1 k2)
// (compile-time constants), A-point transforms over n1 for each k2; the result is put back in natural order
// (all indices are compile-time: the final permutation is a renaming of registers).  A stage built on these
// moves a point through LDS once where two radix-A / radix-B stages move it twice.
__device__ __forceinline__ float2 cmul_const(float2 a, float cr, float ci)
{
    OIP_FFT_FMA
    return make_float2(a.x * cr - a.y * ci, a.x * ci + a.y * cr);
}
template <int A, int B> struct CompositeTw;
template <> struct CompositeTw<5, 5> {
    static __device__ __forceinline__ float2 w(int t)
    {
        constexpr float re[25] = {1.0f, 0.968583161f, 0.87630668f, 0.728968627f, 0.535826795f, 0.309016994f, 0.0627905195f, -0.187381315f, -0.425779292f, -0.63742399f, -0.809016994f, -0.929776486f, -0.992114701f, -0.992114701f, -0.929776486f, -0.809016994f, -0.63742399f, -0.425779292f, -0.187381315f, 0.0627905195f, 0.309016994f, 0.535826795f, 0.728968627f, 0.87630668f, 0.968583161f};
        constexpr float im[25] = {-0.0f, -0.248689887f, -0.481753674f, -0.684547106f, -0.844327926f, -0.951056516f, -0.998026728f, -0.982287251f, -0.904827052f, -0.770513243f, -0.587785252f, -0.368124553f, -0.125333234f, 0.125333234f, 0.368124553f, 0.587785252f, 0.770513243f, 0.904827052f, 0.982287251f, 0.998026728f, 0.951056516f, 0.844327926f, 0.684547106f, 0.481753674f, 0.248689887f};
        return make_float2(re[t], im[t]);
    }
};
template <> struct CompositeTw<3, 5> {
    static __device__ __forceinline__ float2 w(int t)
    {
        constexpr float re[15] = {1.0f, 0.913545458f, 0.669130606f, 0.309016994f, -0.104528463f, -0.5f, -0.809016994f, -0.978147601f, -0.978147601f, -0.809016994f, -0.5f, -0.104528463f, 0.309016994f, 0.669130606f, 0.913545458f};
        constexpr float im[15] = {-0.0f, -0.406736643f, -0.743144825f, -0.951056516f, -0.994521895f, -0.866025404f, -0.587785252f, -0.207911691f, 0.207911691f, 0.587785252f, 0.866025404f, 0.994521895f, 0.951056516f, 0.743144825f, 0.406736643f};
        return make_float2(re[t], im[t]);
    }
};
template <int A, int B> __device__ __forceinline__ void butterfly_leaf(float2 *x);
template <int A, int B> __device__ __forceinline__ void bf_composite(float2 *x)
{
    constexpr int R = A * B;
    // B-point transforms over n2, for each n1
#pragma unroll
    for (int n1 = 0; n1 < A; ++n1) {
        float2 t[B];
#pragma unroll
        for (int n2 = 0; n2 < B; ++n2) t[n2] = x[n1 + A * n2];
        butterfly_leaf<B, 0>(t);
#pragma unroll
        for (int k2 = 0; k2 < B; ++k2) {
            float2 v = t[k2];
            if (n1 * k2 != 0) { const float2 c = CompositeTw<A, B>::w(n1 * k2); v = cmul_const(v, c.x, c.y); }
            x[n1 + A * k2] = v;
        }
    }
    // A-point transforms over n1, for each k2; X[B k1 + k2] lands at position k1 + A k2
    float2 y[R];
#pragma unroll
    for (int k2 = 0; k2 < B; ++k2) {
        float2 u[A];
#pragma unroll
        for (int n1 = 0; n1 < A; ++n1) u[n1] = x[n1 + A * k2];
        butterfly_leaf<A, 0>(u);
#pragma unroll
        for (int k1 = 0; k1 < A; ++k1) y[B * k1 + k2] = u[k1];
    }
#pragma unroll
    for (int m = 0; m < R; ++m) x[m] = y[m];
}
template <int A, int B> __device__ __forceinline__ void butterfly_leaf(float2 *x)
{
    if (A == 2) bf2(x);
    else if (A == 3) bf3(x);
    else if (A == 4) bf4(x);
    else if (A == 5) bf5(x);
    else bf8(x);
}
template <int R> __device__ __forceinline__ void butterfly(float2 *x)
{
    if (R == 2) bf2(x);
    else if (R == 3) bf3(x);
    else if (R == 4) bf4(x);
    else if (R == 5) bf5(x);
    else if (R == 15) bf_composite<3, 5>(x);
    else if (R == 25) bf_composite<5, 5>(x);
    else bf8(x);
}

template <int... Rs> struct RadixList {};

template <int F, int First, int... Rest> struct TwTable {          // entries of w_F^t the stages need
    static constexpr int rest_max(int acc) { return acc; }
    static constexpr int value()
    {
        int m = 1;
        const int r[] = {Rest..., 0};
        for (int i = 0; r[i]; ++i) m = (F / r[i]) > m ? (F / r[i]) : m;
        return m;
    }
};

template <int F, int VS, int VP, int NT, int Ns, int R>
__device__ __forceinline__ void stage_ct(float2 *__restrict__ buf, const float2 *__restrict__ tw, int tid)
{
    constexpr int kFftBlock = NT;
    constexpr int V = 1 << VS;
    constexpr int Vp = VP;              // LDS pitch between consecutive points (>= V)
    constexpr int NB = F / R;
    constexpr int ITEMS = NB << VS;
    constexpr int PER = (ITEMS + kFftBlock - 1) / kFftBlock;
    constexpr int TWSTEP = F / (Ns * R);
    float2 x[PER][R];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int item = tid + i * kFftBlock;
        if (ITEMS % kFftBlock == 0 || item < ITEMS) {
            const int v = item & (V - 1), b = item >> VS;
#pragma unroll
            for (int m = 0; m < R; ++m) x[i][m] = buf[(b + m * NB) * Vp + v];
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int item = tid + i * kFftBlock;
        if (ITEMS % kFftBlock == 0 || item < ITEMS) {
            const int v = item & (V - 1), b = item >> VS;
            const int k = b % Ns;
            if (Ns > 1) {
                // w^(k m) from w^k by repeated products: one table look-up per butterfly
                const float2 w1 = tw[k * TWSTEP];
                float2 w = w1;
#pragma unroll
                for (int m = 1; m < R; ++m) {
                    x[i][m] = cmul(x[i][m], w);
                    if (m + 1 < R) w = cmul(w, w1);
                }
            }
            butterfly<R>(x[i]);
            const int j0 = (b - k) * R + k;
#pragma unroll
            for (int m = 0; m < R; ++m) buf[(j0 + m * Ns) * Vp + v] = x[i][m];
        }
    }
    __syncthreads();
}

template <int F, int VS, int VP, int NT, int Ns, int... Rs> struct Stages;
template <int F, int VS, int VP, int NT, int Ns> struct Stages<F, VS, VP, NT, Ns> {
    static __device__ __forceinline__ void run(float2 *, const float2 *, int = 0) {}
};
template <int F, int VS, int VP, int NT, int Ns, int R, int... Rest> struct Stages<F, VS, VP, NT, Ns, R, Rest...> {
    // tid: threadIdx.x, or an opaque copy of it inside persistent loops (keeps the stage addressing from
    // being hoisted out of the loop and held in registers)
    static __device__ __forceinline__ void run(float2 *buf, const float2 *tw, int tid = threadIdx.x)
    {
        stage_ct<F, VS, VP, NT, Ns, R>(buf, tw, tid);
        Stages<F, VS, VP, NT, Ns * R, Rest...>::run(buf, tw, tid);
    }
};


// One Stockham stage over NA independent two-line buffers (buffer a at buf + a * 2F), software
// pipelined across the buffers: while buffer a is being combined and stored, the loads of buffer
// a+1 are already issued.  One barrier per buffer and stage; only two buffers' worth of points live
// in registers.  Entry: all writers of the buffers are behind a barrier.  Exit: the stores of the
// LAST buffer are not yet behind a barrier (the next stage starts on buffer 0, whose stores are).
template <int F, int NT, int Ns, int R> struct StageOps {
    static constexpr int NB = F / R;
    static constexpr int ITEMS = NB * 2;
    static constexpr int PER = (ITEMS + NT - 1) / NT;
    static constexpr int TWSTEP = F / (Ns * R);
    static __device__ __forceinline__ void load(const float2 *__restrict__ buf, float2 (&x)[PER][R], int tid)
    {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int item = tid + i * NT;
            if (ITEMS % NT == 0 || item < ITEMS) {
                const int v = item & 1, b = item >> 1;
#pragma unroll
                for (int m = 0; m < R; ++m) x[i][m] = buf[(b + m * NB) * 2 + v];
            }
        }
    }
    static __device__ __forceinline__ void store(float2 *__restrict__ buf, const float2 *__restrict__ tw, float2 (&x)[PER][R], int tid)
    {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int item = tid + i * NT;
            if (ITEMS % NT == 0 || item < ITEMS) {
                const int v = item & 1, b = item >> 1;
                const int k = b % Ns;
                if (Ns > 1) {
                    const float2 w1 = tw[k * TWSTEP];
                    float2 w = w1;
#pragma unroll
                    for (int m = 1; m < R; ++m) {
                        x[i][m] = cmul(x[i][m], w);
                        if (m + 1 < R) w = cmul(w, w1);
                    }
                }
                butterfly<R>(x[i]);
                const int j0 = (b - k) * R + k;
#pragma unroll
                for (int m = 0; m < R; ++m) buf[(j0 + m * Ns) * 2 + v] = x[i][m];
            }
        }
    }
};

// The same stage with one item = one butterfly of BOTH lines of a two-line buffer ([point][line] interleave:
// a point's two lines are one 16-byte word).  The twiddle look-up, the power chain and the address arithmetic
// are shared by the two butterflies and the LDS accesses are 16 bytes wide -- about a quarter fewer
// instructions per butterfly; it pays wherever there are enough butterflies to keep the block busy
// (kPairStage below).
template <int F, int NT, int Ns, int R> struct StageOpsPair {
    static constexpr int NB = F / R;
    static constexpr int ITEMS = NB;
    static constexpr int PER = (ITEMS + NT - 1) / NT;
    static constexpr int TWSTEP = F / (Ns * R);
    static constexpr int XR = 2 * R;                    // registers per item: line 0 points, then line 1 points
    static __device__ __forceinline__ void load(const float2 *__restrict__ buf, float2 (&x)[PER][XR], int tid)
    {
        const float4 *__restrict__ p4 = reinterpret_cast<const float4 *>(buf);
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int b = tid + i * NT;
            if (ITEMS % NT == 0 || b < ITEMS) {
#pragma unroll
                for (int m = 0; m < R; ++m) {
                    const float4 v = p4[b + m * NB];
                    x[i][m] = make_float2(v.x, v.y);
                    x[i][R + m] = make_float2(v.z, v.w);
                }
            }
        }
    }
    static __device__ __forceinline__ void store(float2 *__restrict__ buf, const float2 *__restrict__ tw, float2 (&x)[PER][XR], int tid)
    {
        float4 *__restrict__ p4 = reinterpret_cast<float4 *>(buf);
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int b = tid + i * NT;
            if (ITEMS % NT == 0 || b < ITEMS) {
                const int k = b % Ns;
                if (Ns > 1) {
                    const float2 w1 = tw[k * TWSTEP];
                    float2 w = w1;
#pragma unroll
                    for (int m = 1; m < R; ++m) {
                        x[i][m] = cmul(x[i][m], w);
                        x[i][R + m] = cmul(x[i][R + m], w);
                        if (m + 1 < R) w = cmul(w, w1);
                    }
                }
                butterfly<R>(x[i]);
                butterfly<R>(x[i] + R);
                const int j0 = (b - k) * R + k;
#pragma unroll
                for (int m = 0; m < R; ++m) p4[j0 + m * Ns] = make_float4(x[i][m].x, x[i][m].y, x[i][R + m].x, x[i][R + m].y);
            }
        }
    }
};

// pair items when they still fill the block, or when single items would need three rounds
template <int F, int NT, int R> struct kPairStage {
    static constexpr bool value = R >= 3 && ((F / R >= NT * 3 / 4) || ((2 * (F / R) + NT - 1) / NT >= 3));
};
template <int F, int NT, int Ns, int R, bool PAIR> struct StageSelect { using type = StageOps<F, NT, Ns, R>; static constexpr int XR = R; };
template <int F, int NT, int Ns, int R> struct StageSelect<F, NT, Ns, R, true> { using type = StageOpsPair<F, NT, Ns, R>; static constexpr int XR = 2 * R; };

template <int F, int NT, int NA, int Ns, int... Rs> struct StagesPipe;
template <int F, int NT, int NA, int Ns> struct StagesPipe<F, NT, NA, Ns> {
    static __device__ __forceinline__ void run(float2 *, const float2 *, int) { if (NA > 1) __syncthreads(); }
};
template <int F, int NT, int NA, int Ns, int R, int... Rest> struct StagesPipe<F, NT, NA, Ns, R, Rest...> {
    static __device__ __forceinline__ void run(float2 *buf, const float2 *tw, int tid)
    {
        using Sel = StageSelect<F, NT, Ns, R, kPairStage<F, NT, R>::value>;
        using Ops = typename Sel::type;
        float2 x[2][Ops::PER][Sel::XR];
        Ops::load(buf, x[0], tid);
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            __syncthreads();
            if (a + 1 < NA) Ops::load(buf + (a + 1) * 2 * F, x[(a + 1) & 1], tid);
            Ops::store(buf + a * 2 * F, tw, x[a & 1], tid);
        }
        if (NA == 1) __syncthreads();
        StagesPipe<F, NT, NA, Ns * R, Rest...>::run(buf, tw, tid);
    }
};

// One Stockham stage over ALL NA two-line buffers at once: an item is one butterfly of one line of one buffer, items are
// dealt over the whole block (with radix-25 / radix-15 butterflies one buffer alone would keep a third of the block
// busy); load everything, barrier, combine and store, barrier.  Twiddle powers w^m come from w^1 by products, grouped
// as w^(n1) (w^A)^(n2) for composite radices so the chain stays four deep.
template <int F, int NT, int NA, int Ns, int R> struct StageAllOps {
    static constexpr int NB = F / R;
    static constexpr bool POW2 = (NA & (NA - 1)) == 0;
    // power-of-two buffer counts: item bits are  line | butterfly low 3 bits | buffer | butterfly high bits  (no division: a
    // quarter of the row stage's vector instructions was integer arithmetic).  The 16 lanes that share a ds_write_b64 group
    // then belong to ONE buffer: their stores (8 consecutive butterflies x 2 lines) hit 16 different even banks.  With the
    // buffer bit below the butterfly bits (round 2) the two buffers of a group -- 2 F float2 = a multiple of 32 dwords apart --
    // collided on every store (SQ_LDS_BANK_CONFLICT 37 % of the LDS cycles, profiles/r03_pmc_passes.json).  The 32 lanes of a
    // ds_read_b64 group cover both buffers, which sit half the 64 banks apart.  Butterflies are dealt in blocks of 8, so a
    // butterfly count that is not a multiple of 8 leaves a few slots empty.
    static constexpr int NB8 = (NB + 7) / 8 * 8;
    static constexpr int ITEMS = POW2 ? NA * 2 * NB8 : NA * 2 * NB;
    static constexpr int PER = (ITEMS + NT - 1) / NT;
    static constexpr int TWSTEP = F / (Ns * R);
    static constexpr int A = R == 15 ? 3 : (R == 25 ? 5 : 1);       // composite split R = A * B (1: plain radix)
    static constexpr bool GUARD = POW2 && NB8 != NB;                // some slots hold no butterfly
    static __device__ __forceinline__ void decode(int item, int *base, int *b)
    {
        if (POW2) {
            constexpr int LA = NA == 1 ? 0 : (NA == 2 ? 1 : (NA == 4 ? 2 : 3));
            *b = ((item >> 1) & 7) | ((item >> (4 + LA)) << 3);
            *base = ((item >> 4) & (NA - 1)) * 2 * F + (item & 1);
            return;
        }
        const int a = item / (2 * NB), rem = item - a * (2 * NB);
        *b = rem >> 1;
        *base = a * 2 * F + (rem & 1);                                // element offset of (buffer a, line), point 0
    }
    static __device__ __forceinline__ void load(const float2 *__restrict__ buf, float2 (&x)[PER][R], int tid)
    {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int item = tid + i * NT;
            if (ITEMS % NT == 0 || item < ITEMS) {
                int base, b;
                decode(item, &base, &b);
                if (GUARD && b >= NB) continue;
#pragma unroll
                for (int m = 0; m < R; ++m) x[i][m] = buf[base + 2 * (b + m * NB)];
            }
        }
    }
    static __device__ __forceinline__ void store(float2 *__restrict__ buf, const float2 *__restrict__ tw, float2 (&x)[PER][R], int tid)
    {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int item = tid + i * NT;
            if (ITEMS % NT == 0 || item < ITEMS) {
                int base, b;
                decode(item, &base, &b);
                if (GUARD && b >= NB) continue;
                const int k = b % Ns;
                if (Ns > 1) {
                    const float2 w1 = tw[k * TWSTEP];
                    if (A == 1) {
                        float2 w = w1;
#pragma unroll
                        for (int m = 1; m < R; ++m) {
                            x[i][m] = cmul(x[i][m], w);
                            if (m + 1 < R) w = cmul(w, w1);
                        }
                    } else {
                        constexpr int B = R / A;
                        float2 wn1[A];                                  // w^n1, n1 < A
                        wn1[0] = make_float2(1.f, 0.f);
                        wn1[1] = w1;
#pragma unroll
                        for (int n1 = 2; n1 < A; ++n1) wn1[n1] = cmul(wn1[n1 - 1], w1);
                        float2 wa = cmul(wn1[A - 1], w1);               // w^A
                        float2 wan = wa;                                // (w^A)^n2
#pragma unroll
                        for (int n1 = 1; n1 < A; ++n1) x[i][n1] = cmul(x[i][n1], wn1[n1]);
#pragma unroll
                        for (int n2 = 1; n2 < B; ++n2) {
                            x[i][A * n2] = cmul(x[i][A * n2], wan);
#pragma unroll
                            for (int n1 = 1; n1 < A; ++n1) x[i][n1 + A * n2] = cmul(x[i][n1 + A * n2], cmul(wn1[n1], wan));
                            if (n2 + 1 < B) wan = cmul(wan, wa);
                        }
                    }
                }
                butterfly<R>(x[i]);
                const int j0 = (b - k) * R + k;
#pragma unroll
                for (int m = 0; m < R; ++m) buf[base + 2 * (j0 + m * Ns)] = x[i][m];
            }
        }
    }
};
template <int F, int NT, int NA, int Ns, int... Rs> struct StagesAll;
template <int F, int NT, int NA, int Ns> struct StagesAll<F, NT, NA, Ns> {
    static __device__ __forceinline__ void run(float2 *, const float2 *, int) {}
};
template <int F, int NT, int NA, int Ns, int R, int... Rest> struct StagesAll<F, NT, NA, Ns, R, Rest...> {
    // entry: every writer of the buffers is behind a barrier; exit: likewise
    static __device__ __forceinline__ void run(float2 *buf, const float2 *tw, int tid)
    {
        using Ops = StageAllOps<F, NT, NA, Ns, R>;
        {
            float2 x[Ops::PER][R];
            Ops::load(buf, x, tid);
            __syncthreads();
            Ops::store(buf, tw, x, tid);
        }
        __syncthreads();
        StagesAll<F, NT, NA, Ns * R, Rest...>::run(buf, tw, tid);
    }
};

// Two independent three-stage transforms (buffers bp: NAP x two lines of FP points, bn: NAN_ x two lines of FN points)
// interleaved: between two barriers one stores a stage and the other loads its next one, so the six stages cost seven
// barriers instead of twelve while only one register array is live at a time.
template <class OpsA, class OpsB, int RA, int RB>
__device__ __forceinline__ void stage_store_then_load(float2 *ba, const float2 *twa, float2 (&xa)[OpsA::PER][RA], float2 *bb,
                                                      float2 (&xb)[OpsB::PER][RB], int tid)
{
    OpsA::store(ba, twa, xa, tid);
    __builtin_amdgcn_sched_barrier(0);
    OpsB::load(bb, xb, tid);
}
template <int FP, int NAP, int FN, int NAN_, int NT, int R1, int R2, int R3P, int R3N> struct StagesDual3 {
    static __device__ __forceinline__ void run(float2 *bp, const float2 *twp, float2 *bn, const float2 *twn, int tid)
    {
        using P1 = StageAllOps<FP, NT, NAP, 1, R1>;        using N1 = StageAllOps<FN, NT, NAN_, 1, R1>;
        using P2 = StageAllOps<FP, NT, NAP, R1, R2>;       using N2 = StageAllOps<FN, NT, NAN_, R1, R2>;
        using P3 = StageAllOps<FP, NT, NAP, R1 * R2, R3P>; using N3 = StageAllOps<FN, NT, NAN_, R1 * R2, R3N>;
        float2 a1[P1::PER][R1];
        P1::load(bp, a1, tid);
        __syncthreads();
        float2 b1[N1::PER][R1];
        stage_store_then_load<P1, N1, R1, R1>(bp, twp, a1, bn, b1, tid);
        __syncthreads();
        float2 a2[P2::PER][R2];
        stage_store_then_load<N1, P2, R1, R2>(bn, twn, b1, bp, a2, tid);
        __syncthreads();
        float2 b2[N2::PER][R2];
        stage_store_then_load<P2, N2, R2, R2>(bp, twp, a2, bn, b2, tid);
        __syncthreads();
        float2 a3[P3::PER][R3P];
        stage_store_then_load<N2, P3, R2, R3P>(bn, twn, b2, bp, a3, tid);
        __syncthreads();
        float2 b3[N3::PER][R3N];
        stage_store_then_load<P3, N3, R3P, R3N>(bp, twp, a3, bn, b3, tid);
        __syncthreads();
        N3::store(bn, twn, b3, tid);
        __syncthreads();
    }
};

}  // namespace oipfft
