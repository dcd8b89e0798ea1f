// ubench_valu.hip -- issue rate of integer vector instructions on gfx950.
//
// Every wavefront runs a long stream of independent instructions of one kind;
// the grid puts 1, 2, 4 or 8 wavefronts on every SIMD.  Prints cycles per
// wave-instruction and SIMD (s_memtime ticks of the slowest wavefront divided
// by the instructions all wavefronts of a SIMD issued).
//
//   hipcc --offload-arch=gfx950 -O2 -o ubench_valu ubench_valu.hip && ./ubench_valu
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP32(x) REP8(x) REP8(x) REP8(x) REP8(x)

constexpr int ITERS = 2000;

#define KERNEL(name, body)                                                                          \
	__global__ __launch_bounds__(256) void name(uint32_t *out, uint64_t *ticks)                 \
	{                                                                                           \
		uint32_t a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 9, a5 = a0 ^ 77, \
			 a6 = a0 + 1, a7 = a0 + 2;                                                  \
		uint32_t b = blockIdx.x + 1, c = threadIdx.x | 1;                                   \
		const uint64_t t0 = __builtin_amdgcn_s_memtime();                                   \
		for (int i = 0; i < ITERS; i++) {                                                   \
			asm volatile(REP8(body)                                                     \
				     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), \
				       "+v"(a7)                                                     \
				     : "v"(b), "v"(c)                                               \
				     : "vcc");                                                      \
		}                                                                                   \
		const uint64_t t1 = __builtin_amdgcn_s_memtime();                                   \
		out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7; \
		if ((threadIdx.x & 63) == 0)                                                        \
			ticks[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;              \
	}

// eight independent instructions per body (one per accumulator)
#define OP8(fmt)                                                                                    \
	fmt("%0") fmt("%1") fmt("%2") fmt("%3") fmt("%4") fmt("%5") fmt("%6") fmt("%7")

#define F_ADD(r) "v_add_u32 " r ", " r ", %8\n"
#define F_LSHLOR(r) "v_lshl_or_b32 " r ", " r ", 3, %8\n"
#define F_BFE(r) "v_bfe_u32 " r ", " r ", 3, 29\n"
#define F_FFBH(r) "v_ffbh_u32 " r ", " r "\n"
#define F_PERM(r) "v_perm_b32 " r ", " r ", %8, %9\n"
#define F_DPP(r) "v_mov_b32_dpp " r ", " r " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define F_ADDDPP(r) "v_add_u32_dpp " r ", " r ", " r " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define F_MIN(r) "v_min_u32 " r ", " r ", %8\n"
#define F_MUL24(r) "v_mul_u32_u24 " r ", " r ", %8\n"
#define F_MULLO(r) "v_mul_lo_u32 " r ", " r ", %8\n"
#define F_ANDOR(r) "v_and_or_b32 " r ", " r ", %8, %9\n"
#define F_ALIGN(r) "v_alignbit_b32 " r ", " r ", %8, 5\n"
#define F_CMPSEL(r) "v_cmp_lt_u32 vcc, " r ", %8\n v_cndmask_b32 " r ", " r ", %9, vcc\n"
#define F_ADDCO(r) "v_add_co_u32 " r ", vcc, " r ", %8\n"
#define F_PKADD(r) "v_pk_add_u16 " r ", " r ", %8\n"
#define F_PKMIN(r) "v_pk_min_u16 " r ", " r ", %8\n"
#define F_SDWA(r) "v_add_u32_sdwa " r ", " r ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
#define F_FMA(r) "v_fma_f32 " r ", " r ", %8, %9\n"
#define F_MAD24(r) "v_mad_u32_u24 " r ", " r ", %8, %9\n"
#define F_ADD3(r) "v_add3_u32 " r ", " r ", %8, %9\n"
#define F_LSHLADD(r) "v_lshl_add_u32 " r ", " r ", 2, %8\n"
#define F_XOR(r) "v_xor_b32 " r ", " r ", %8\n"
#define F_SUBREV(r) "v_subrev_u32 " r ", " r ", %8\n"
#define F_MED3(r) "v_med3_u32 " r ", " r ", %8, %9\n"
#define F_BCNT(r) "v_bcnt_u32_b32 " r ", " r ", %8\n"
#define F_MBCNT(r) "v_mbcnt_lo_u32_b32 " r ", " r ", %8\n"
#define F_SAD(r) "v_sad_u8 " r ", " r ", %8, %9\n"
#define F_LSHR64(r) "v_lshrrev_b32 " r ", 3, " r "\n"


#define F_SUB(r) "v_sub_u32 " r ", " r ", %8\n"
#define F_AND(r) "v_and_b32 " r ", " r ", %8\n"
#define F_OR(r) "v_or_b32 " r ", " r ", %8\n"
#define F_LSHL(r) "v_lshlrev_b32 " r ", 3, " r "\n"
#define F_LSHLV(r) "v_lshlrev_b32 " r ", %8, " r "\n"
#define F_ASHR(r) "v_ashrrev_i32 " r ", 3, " r "\n"
#define F_MAXU(r) "v_max_u32 " r ", " r ", %8\n"
#define F_MINI(r) "v_min_i32 " r ", " r ", %8\n"
#define F_MOV(r) "v_mov_b32 " r ", %8\n"
#define F_CND(r) "v_cndmask_b32 " r ", " r ", %9, vcc\n"
#define F_CMP(r) "v_cmp_lt_u32 vcc, " r ", %8\n"
#define F_NOT(r) "v_not_b32 " r ", " r "\n"
#define F_ANDLIT(r) "v_and_b32 " r ", 0x12345678, " r "\n"
#define F_ADDLIT(r) "v_add_u32 " r ", 0x12345678, " r "\n"
#define F_ADDF(r) "v_add_f32 " r ", " r ", %8\n"
#define F_MULF(r) "v_mul_f32 " r ", " r ", %8\n"
#define F_MAXF(r) "v_max_f32 " r ", " r ", %8\n"
#define F_MINF(r) "v_min_f32 " r ", " r ", %8\n"
#define F_CVTFU(r) "v_cvt_f32_u32 " r ", " r "\n"
#define F_CVTUB(r) "v_cvt_f32_ubyte1 " r ", " r "\n"
#define F_FREXP(r) "v_frexp_exp_i32_f32 " r ", " r "\n"
#define F_FMAC(r) "v_fmac_f32 " r ", %8, %9\n"
#define F_XNOR(r) "v_xnor_b32 " r ", " r ", %8\n"
#define F_SUBB(r) "v_subb_co_u32 " r ", vcc, " r ", %8, vcc\n"
#define F_ADDSGPR(r) "v_add_u32 " r ", s4, " r "\n"
#define F_READFL(r) "v_readfirstlane_b32 s4, " r "\n"
#define F_ADD_OR(r) "v_add_u32 " r ", " r ", %8\n v_or_b32 " r ", " r ", %9\n"
#define F_ADD_BFE(r) "v_add_u32 " r ", " r ", %8\n v_bfe_u32 " r ", " r ", 3, 29\n"
#define F_LSHL64(r) "v_lshlrev_b64 v[100:101], 3, v[100:101]\n"
#define F_DSADD(r) "ds_add_u32 " r ", %8\n"
#define F_BPERM(r) "ds_bpermute_b32 " r ", %9, " r "\n s_waitcnt lgkmcnt(0)\n"
#define F_OR3(r) "v_or3_b32 " r ", " r ", %8, %9\n"
#define F_BFI(r) "v_bfi_b32 " r ", " r ", %8, %9\n"
#define F_MAX3(r) "v_max3_u32 " r ", " r ", %8, %9\n"
#define F_LSHLREV16(r) "v_lshlrev_b16 " r ", 3, " r "\n"
#define F_ADDU16(r) "v_add_u16 " r ", " r ", %8\n"
#define F_SUBBREV(r) "v_subrev_u32 " r ", " r ", %8\n"

KERNEL(k_add, OP8(F_ADD))
KERNEL(k_lshlor, OP8(F_LSHLOR))
KERNEL(k_bfe, OP8(F_BFE))
KERNEL(k_ffbh, OP8(F_FFBH))
KERNEL(k_perm, OP8(F_PERM))
KERNEL(k_dpp, OP8(F_DPP))
KERNEL(k_adddpp, OP8(F_ADDDPP))
KERNEL(k_min, OP8(F_MIN))
KERNEL(k_mul24, OP8(F_MUL24))
KERNEL(k_mullo, OP8(F_MULLO))
KERNEL(k_andor, OP8(F_ANDOR))
KERNEL(k_align, OP8(F_ALIGN))
KERNEL(k_cmpsel, OP8(F_CMPSEL))
KERNEL(k_addco, OP8(F_ADDCO))
KERNEL(k_pkadd, OP8(F_PKADD))
KERNEL(k_pkmin, OP8(F_PKMIN))
KERNEL(k_sdwa, OP8(F_SDWA))
KERNEL(k_fma, OP8(F_FMA))
KERNEL(k_mad24, OP8(F_MAD24))
KERNEL(k_add3, OP8(F_ADD3))
KERNEL(k_lshladd, OP8(F_LSHLADD))
KERNEL(k_xor, OP8(F_XOR))
KERNEL(k_med3, OP8(F_MED3))
KERNEL(k_bcnt, OP8(F_BCNT))
KERNEL(k_sad, OP8(F_SAD))
KERNEL(k_lshr, OP8(F_LSHR64))


KERNEL(k_sub, OP8(F_SUB))
KERNEL(k_and, OP8(F_AND))
KERNEL(k_or, OP8(F_OR))
KERNEL(k_lshl, OP8(F_LSHL))
KERNEL(k_lshlv, OP8(F_LSHLV))
KERNEL(k_ashr, OP8(F_ASHR))
KERNEL(k_maxu, OP8(F_MAXU))
KERNEL(k_mini, OP8(F_MINI))
KERNEL(k_mov, OP8(F_MOV))
KERNEL(k_cnd, OP8(F_CND))
KERNEL(k_cmp, OP8(F_CMP))
KERNEL(k_not, OP8(F_NOT))
KERNEL(k_andlit, OP8(F_ANDLIT))
KERNEL(k_addlit, OP8(F_ADDLIT))
KERNEL(k_addf, OP8(F_ADDF))
KERNEL(k_mulf, OP8(F_MULF))
KERNEL(k_maxf, OP8(F_MAXF))
KERNEL(k_minf, OP8(F_MINF))
KERNEL(k_cvtfu, OP8(F_CVTFU))
KERNEL(k_cvtub, OP8(F_CVTUB))
KERNEL(k_frexp, OP8(F_FREXP))
KERNEL(k_fmac, OP8(F_FMAC))
KERNEL(k_xnor, OP8(F_XNOR))
KERNEL(k_addsgpr, OP8(F_ADDSGPR))
KERNEL(k_add_or, OP8(F_ADD_OR))
KERNEL(k_add_bfe, OP8(F_ADD_BFE))
KERNEL(k_or3, OP8(F_OR3))
KERNEL(k_bfi, OP8(F_BFI))
KERNEL(k_max3, OP8(F_MAX3))
KERNEL(k_addu16, OP8(F_ADDU16))
KERNEL(k_subrev, OP8(F_SUBBREV))

typedef void (*kern_t)(uint32_t *, uint64_t *);

int main()
{
	struct
	{
		const char *name;
		kern_t k;
		int per_body; // instructions per accumulator and body
	} tests[] = {
		{"v_add_u32", k_add, 1},        {"v_lshl_or_b32", k_lshlor, 1},   {"v_bfe_u32", k_bfe, 1},
		{"v_ffbh_u32", k_ffbh, 1},      {"v_perm_b32", k_perm, 1},        {"v_mov_dpp", k_dpp, 1},
		{"v_add_u32_dpp", k_adddpp, 1}, {"v_min_u32", k_min, 1},          {"v_mul_u32_u24", k_mul24, 1},
		{"v_mul_lo_u32", k_mullo, 1},   {"v_and_or_b32", k_andor, 1},     {"v_alignbit_b32", k_align, 1},
		{"v_cmp+v_cndmask", k_cmpsel, 2}, {"v_add_co_u32", k_addco, 1},   {"v_pk_add_u16", k_pkadd, 1},
		{"v_pk_min_u16", k_pkmin, 1},   {"v_add_u32_sdwa", k_sdwa, 1},    {"v_fma_f32", k_fma, 1},
		{"v_mad_u32_u24", k_mad24, 1},  {"v_add3_u32", k_add3, 1},        {"v_lshl_add_u32", k_lshladd, 1},
		{"v_xor_b32", k_xor, 1},        {"v_med3_u32", k_med3, 1},        {"v_bcnt_u32_b32", k_bcnt, 1},
		{"v_sad_u8", k_sad, 1},         {"v_lshrrev_b32", k_lshr, 1},
		{"v_sub_u32", k_sub, 1}, {"v_and_b32", k_and, 1}, {"v_or_b32", k_or, 1}, {"v_lshlrev_b32 imm", k_lshl, 1},
		{"v_lshlrev_b32 vgpr", k_lshlv, 1}, {"v_ashrrev_i32", k_ashr, 1}, {"v_max_u32", k_maxu, 1}, {"v_min_i32", k_mini, 1},
		{"v_mov_b32", k_mov, 1}, {"v_cndmask_b32", k_cnd, 1}, {"v_cmp_lt_u32", k_cmp, 1}, {"v_not_b32", k_not, 1},
		{"v_and_b32 literal", k_andlit, 1}, {"v_add_u32 literal", k_addlit, 1}, {"v_add_f32", k_addf, 1}, {"v_mul_f32", k_mulf, 1},
		{"v_max_f32", k_maxf, 1}, {"v_min_f32", k_minf, 1}, {"v_cvt_f32_u32", k_cvtfu, 1}, {"v_cvt_f32_ubyte1", k_cvtub, 1},
		{"v_frexp_exp_i32", k_frexp, 1}, {"v_fmac_f32", k_fmac, 1}, {"v_xnor_b32", k_xnor, 1}, {"v_add_u32 sgpr", k_addsgpr, 1},
		{"add+or (2)", k_add_or, 2}, {"add+bfe (2)", k_add_bfe, 2}, {"v_or3_b32", k_or3, 1}, {"v_bfi_b32", k_bfi, 1},
		{"v_max3_u32", k_max3, 1}, {"v_add_u16", k_addu16, 1}, {"v_subrev_u32", k_subrev, 1},
	};
	int cus = 0;
	hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
	uint32_t *out;
	uint64_t *ticks;
	const int max_wgs = cus * 8;
	hipMalloc(&out, size_t(max_wgs) * 256 * 4);
	hipMalloc(&ticks, size_t(max_wgs) * 4 * 8);
	std::vector<uint64_t> h(size_t(max_wgs) * 4);
	printf("%-18s %8s %8s %8s %8s   (cycles per wave-instruction and SIMD; wall-derived GHz-free)\n", "instruction",
	       "1w/SIMD", "2w/SIMD", "4w/SIMD", "8w/SIMD");
	for (auto &t : tests) {
		printf("%-18s", t.name);
		for (int wps : {1, 2, 4, 8}) {
			const int wgs = cus * wps; // one 256-thread workgroup = one wavefront per SIMD
			hipLaunchKernelGGL(t.k, dim3(wgs), dim3(256), 0, 0, out, ticks);
			hipDeviceSynchronize();
			hipEvent_t e0, e1;
			hipEventCreate(&e0);
			hipEventCreate(&e1);
			hipEventRecord(e0, 0);
			hipLaunchKernelGGL(t.k, dim3(wgs), dim3(256), 0, 0, out, ticks);
			hipEventRecord(e1, 0);
			hipDeviceSynchronize();
			float ms = 0;
			hipEventElapsedTime(&ms, e0, e1);
			hipMemcpy(h.data(), ticks, size_t(wgs) * 4 * 8, hipMemcpyDeviceToHost);
			uint64_t mx = 0;
			for (int i = 0; i < wgs * 4; i++)
				mx = h[i] > mx ? h[i] : mx;
			const double instr = double(ITERS) * 64.0 * t.per_body * wps; // per SIMD
			// s_memtime ticks are shader cycles; wall time x an assumed 2.4 GHz is printed too
			printf(" %5.2f/%4.2f", double(mx) / instr, double(ms) * 1e-3 * 2.4e9 / instr);
			hipEventDestroy(e0);
			hipEventDestroy(e1);
		}
		printf("\n");
	}
	return 0;
}
