// Checks the operand-modifier semantics of the packed-f32 inline assembly helpers (dev tool).
#include <hip/hip_runtime.h>
#include <cstdio>
namespace dsa { inline int fail(int, const char*, ...) { return -1; } }
typedef float sf32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ sf32x2 cmul2(sf32x2 a, sf32x2 t)
{
    sf32x2 t1, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t1) : "v"(a), "v"(t));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(r) : "v"(a), "v"(t), "v"(t1));
    return r;
}
__device__ __forceinline__ sf32x2 add_negi(sf32x2 a, sf32x2 b)
{
    sf32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ sf32x2 add_posi(sf32x2 a, sf32x2 b)
{
    sf32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ sf32x2 diff_sum(sf32x2 b)
{
    sf32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,1]" : "=v"(r) : "v"(b), "v"(b));
    return r;
}
__global__ void k(float* o, const float* in)
{
    sf32x2 a = {in[0], in[1]}, b = {in[2], in[3]};
    sf32x2 r0 = cmul2(a, b), r1 = add_negi(a, b), r2 = add_posi(a, b), r3 = diff_sum(a);
    o[0] = r0.x; o[1] = r0.y; o[2] = r1.x; o[3] = r1.y; o[4] = r2.x; o[5] = r2.y; o[6] = r3.x; o[7] = r3.y;
}
int main()
{
    float h[4] = {2.f, 3.f, 5.f, 7.f}, *d, *o, r[8];
    hipMalloc(&d, 16); hipMalloc(&o, 32);
    hipMemcpy(d, h, 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, d);
    hipMemcpy(r, o, 32, hipMemcpyDeviceToHost);
    printf("cmul   got (%g, %g) want (%g, %g)\n", r[0], r[1], 2.f * 5 - 3 * 7, 3.f * 5 + 2 * 7);
    printf("a-ib   got (%g, %g) want (%g, %g)\n", r[2], r[3], 2.f + 7, 3.f - 5);
    printf("a+ib   got (%g, %g) want (%g, %g)\n", r[4], r[5], 2.f - 7, 3.f + 5);
    printf("d,s    got (%g, %g) want (%g, %g)\n", r[6], r[7], 2.f - 3, 2.f + 3);
    return 0;
}
