// Micro-benchmark: issue rate of vector ALU instructions on gfx950 as a function of
// waves per SIMD.  Prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int OP>
__global__ void __launch_bounds__(256) k(float *out, int iters, float seed)
{
	float a[8]; double b[8];
	for (int i = 0; i < 8; i++) { a[i] = seed + i + threadIdx.x; b[i] = seed + i + threadIdx.x; }
	float m = seed * 0.999f; double md = seed * 0.999;
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int i = 0; i < 8; i++) {
			if (OP == 0) a[i] = __builtin_fmaf(a[i], m, m);
			if (OP == 1) b[i] = __builtin_fma(b[i], md, md);
			if (OP == 2) b[i] = b[i] * md;
			if (OP == 3) b[i] = b[i] + md;
			if (OP == 4) a[i] = a[i] * m;
			if (OP == 5) a[i] = __builtin_fmaxf(a[i], m) ;
			if (OP == 6) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a[i]) : "v"(b[i]));
			if (OP == 7) asm volatile("v_rcp_f64 %0, %1" : "=v"(b[i]) : "v"(b[i]));
			if (OP == 8) asm volatile("v_sqrt_f64 %0, %1" : "=v"(b[i]) : "v"(b[i]));
			if (OP == 9) asm volatile("v_rcp_f32 %0, %1" : "=v"(a[i]) : "v"(a[i]));
		}
	}
	float s = 0; for (int i = 0; i < 8; i++) s += a[i] + (float)b[i];
	if (s == 12345.678f) out[0] = s;
}

template <int OP> void run(const char *name, float *d)
{
	const int iters = 20000;
	for (int wps = 1; wps <= 8; wps *= 2) {
		// 256 CUs, `wps` workgroups of 256 threads per CU -> wps waves per SIMD
		int blocks = 256 * wps;
		hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
		hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 100, 1.0f);
		CHECK(hipDeviceSynchronize());
		CHECK(hipEventRecord(e0));
		hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
		CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
		float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
		double insts_per_simd = (double)iters * 8 * wps;        // wave-instructions issued on one SIMD
		double cycles = ms * 1e-3 * 2.4e9;
		printf("%-14s waves/SIMD=%d  %.2f cycles per wave-instruction per SIMD (%.3f ms)\n", name, wps, cycles / insts_per_simd, ms);
	}
}

int main()
{
	float *d; CHECK(hipMalloc(&d, 4));
	run<0>("v_fma_f32", d); run<4>("v_mul_f32", d); run<5>("v_max_f32", d);
	run<1>("v_fma_f64", d); run<2>("v_mul_f64", d); run<3>("v_add_f64", d);
	run<6>("v_cvt_f32_f64", d); run<7>("v_rcp_f64", d); run<8>("v_sqrt_f64", d); run<9>("v_rcp_f32", d);
	return 0;
}
