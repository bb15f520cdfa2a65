// abi_forms.hip -- which rollout kernel form serves a handle: selection by model and K, names, mppi_set_rollout_variant.
#include "abi_internal.hpp"

using namespace mppi;
using namespace mppi_abi;

namespace mppi_abi {

bool use_mfma(const mppi_handle *h)
{
  if (h->basis || h->variant_pref == 2 || h->variant_pref == 3) return false;
  return h->mfma_ok;
}

// "valu" on a standard shape runs the register/scalar-operand kernel; "valu_lds" forces the generic one
bool use_valu_reg(const mppi_handle *h) { return !h->basis && !use_mfma(h) && h->valu_reg_ok && h->variant_pref != 3; }
int effective_block(const mppi_handle *h)
{
  if (h->block_threads != 0) return h->block_threads;
  const int groups = h->K / kRolloutsPerWave;
  const int cus = h->num_simds / 4;
  // 64-wide nets up to one group per CU: the 4x4x1-MFMA form (rollout_m44.hip: every hidden weight in registers, no hand-over
  // between waves; 6-64x4-4 K=1920 149 us against the oct form's 181, 6-64-64-4 K=4096 70.5 against 101:
  // profiles/r04_d_m44_first.txt); its output layer is a butterfly -- inside the north-star tolerance, not bit-identical
  // ("mfma" asked for explicitly keeps the reference-order forms, like the row-tree form below)
  // (6-64-64-4 also at two groups per CU -- 110 VGPRs, two workgroups fit: K=8192 128 us against the oct form's 166)
  if (groups <= (h->n_hidden == 2 ? 2 : 1) * cus && m44_variant_supported(h->hidden, h->n_hidden) && h->variant_pref != 1) return 944;
  if (groups <= 2 * cus && oct_variant_supported(h->hidden, h->n_hidden)) return 800;
  if (multi_variant_supported(h->hidden, h->n_hidden)) {
    // 6-32-32-4 at one group per CU: the vector-ALU ROW form (rollout_row.hip) -- the shortest recurrence of all
    // (K=4096, T=100: 56.8 us; quad 68.2 us)
    // ("mfma" asked for explicitly -- the A/B arm of SURVEY cfg 4 -- keeps the matrix-instruction forms)
    // -- in its TREE form (901: the output layer as per-lane partials + a butterfly, rollout 45.8 -> 36.7 us; inside the
    // north-star tolerance of the reference's summation order, tests/test_row_tree_gpu.py); "row_exact" keeps the
    // k-ascending output chain (900), bit-identical to every other form
    if (groups <= cus && row_variant_supported(h->hidden, h->n_hidden) && h->variant_pref != 1) return 901;
    if (groups <= cus) return 512;
    if (groups <= 2 * cus) return 1002;
    // (64-wide nets beyond one group per SIMD: the eight-wave form needs 172 VGPRs = one workgroup per CU, so K = 32768
    // runs in two rounds.  The six-wave form "multi4u" -- 168 VGPRs, three waves per SIMD, two workgroups per CU =
    // two dynamics waves + one rider per SIMD -- was measured against it: K=32768, T=150, 6-64-64-4 0.584 ms vs
    // 0.538 ms; K=16384 0.298 vs 0.272 ms.  Two f32-MFMA waves on one SIMD take the sum of their times (the f32
    // MFMA occupies the vector datapath), so co-residence buys nothing and the single cost wave is the
    // slower rider.  Not chosen automatically; kept as an A/B arm.)
    return 1004;
  }
  return (4 * groups <= h->num_simds) ? 512 : 256;
}

// multi form: eps from the stand-alone generator kernel (forced by "_gen", and the automatic choice for ND = 4)
bool multi_gen(const mppi_handle *h)
{
  if (h->block_threads != 0) return h->multi_standalone_noise;
  return effective_block(h) == 1004 || effective_block(h) == 1040;
}

// basis-function model, wavefronts per 64 rollouts: dynamics + cost + control wave (in-kernel generator) while
// each gets a SIMD of its own; dynamics + cost wave ("quad") up to twice that; one wave ("fused" / "block64")
int bf_waves(const mppi_handle *h)
{
  if (h->block_threads == 64 || h->block_threads == 256) return 1;
  if (h->block_threads == 512) return 2;
  if (h->block_threads == 768) return 3;
  if (3 * (h->K / 64) <= h->num_simds) return 3;
  return (2 * (h->K / 64) <= 2 * h->num_simds) ? 2 : 1;
}

// the quad and multi MFMA kernels carry their own control/noise wavefront
bool has_noise_wave(const mppi_handle *h)
{
  if (h->basis) return bf_waves(h) == 3;
  if (!use_mfma(h)) return false;
  const int b = effective_block(h);
  return b == 512 || is_row(b) || is_row64(b) || is_m44(b) || ((b == 800 || b > 1000) && !multi_gen(h));
}

}  // namespace mppi_abi

extern "C" {

const char *mppi_rollout_variant(const mppi_handle *h)
{
  if (!h) return "";
  if (h->basis) return bf_waves(h) == 3 ? "basis_funcs25_valu_3w" : bf_waves(h) == 2 ? "basis_funcs25_valu_2w" : "basis_funcs25_valu";
  if (!use_mfma(h)) return use_valu_reg(h) ? "valu_reg_lds" : "valu_lds";
  static thread_local char buf[64];
  const int b = effective_block(h);
  if (b == 1040)
    snprintf(buf, sizeof(buf), "mfma16x16x4_h%d_l%d_multi4u%s", h->hidden, h->n_hidden, multi_gen(h) ? "_gen" : "");
  else if (b > 1000)
    snprintf(buf, sizeof(buf), "mfma16x16x4_h%d_l%d_multi%d%s", h->hidden, h->n_hidden, b - 1000,
             multi_gen(h) ? "_gen" : "");
  else if (is_row(b))
    snprintf(buf, sizeof(buf), "valu_row8w%s_h%d_l%d", b == 901 ? "_tree" : "", h->hidden, h->n_hidden);
  else if (is_row64(b))
    snprintf(buf, sizeof(buf), "valu_row64_r%d_tree_h%d_l%d", b - 900, h->hidden, h->n_hidden);
  else if (is_m44(b))
    snprintf(buf, sizeof(buf), "mfma4x4x1_h%d_l%d_m44_tree", h->hidden, h->n_hidden);
  else
    snprintf(buf, sizeof(buf), "mfma16x16x4_h%d_l%d_%s", h->hidden, h->n_hidden,
             b == 512 ? "quad4w" : b == 800 ? (multi_gen(h) ? "oct8w_gen" : "oct8w") : (b == 256 ? "fused_b256" : "fused_b64"));
  return buf;
}

int mppi_set_rollout_variant(mppi_handle *h, const char *name)
{
  if (!h || !name) return MPPI_ERR_INVALID;
  if (strcmp(name, "auto") == 0) {
    h->variant_pref = 0;
    h->block_threads = 0;
  }
  else if (strcmp(name, "mfma") == 0) {
    if (!h->mfma_ok) return fail(h, MPPI_ERR_UNSUPPORTED, "MFMA variant needs 6-HxN-4 with H in {32,64}, N in {2,4}");
    h->variant_pref = 1;
  } else if (strcmp(name, "valu") == 0) h->variant_pref = 2;
  else if (strcmp(name, "valu_lds") == 0) h->variant_pref = 3;
  else if (strcmp(name, "quad") == 0) h->block_threads = 512;
  else if (strcmp(name, "bf3") == 0) {
    if (!h->basis) return fail(h, MPPI_ERR_UNSUPPORTED, "bf3 is a form of the basis-function model");
    h->block_threads = 768;
  }
  else if (strcmp(name, "row") == 0 || strcmp(name, "row_exact") == 0 || strcmp(name, "row_tree") == 0) {
    if (!h->mfma_ok || !row_variant_supported(h->hidden, h->n_hidden))
      return fail(h, MPPI_ERR_UNSUPPORTED, "row form exists for 6-32x2-4");
    h->block_threads = strcmp(name, "row_tree") == 0 ? 901 : 900;
  }
  else if (strcmp(name, "m44") == 0) {
    if (!h->mfma_ok || !m44_variant_supported(h->hidden, h->n_hidden))
      return fail(h, MPPI_ERR_UNSUPPORTED, "m44 form exists for 6-64x2-4 and 6-64x4-4");
    h->block_threads = 944;
  }
  else if (strcmp(name, "row64") == 0 || strcmp(name, "row64_r8") == 0 || strcmp(name, "row64_r16") == 0) {
    if (!h->mfma_ok || !row64_variant_supported(h->hidden, h->n_hidden))
      return fail(h, MPPI_ERR_UNSUPPORTED, "row64 form exists for 6-64x2-4 and 6-64x4-4");
    // 8 rollouts per group (one dynamics wave per SIMD) while every such group has a CU of its own, else 16
    const int r = name[5] == 0 ? ((h->K / 8 <= h->num_simds / 4) ? 8 : 16) : (name[7] == '8' ? 8 : 16);
    h->block_threads = 900 + r;
  }
  else if (strcmp(name, "oct") == 0 || strcmp(name, "oct_gen") == 0) {
    if (!h->mfma_ok || !oct_variant_supported(h->hidden, h->n_hidden))
      return fail(h, MPPI_ERR_UNSUPPORTED, "oct form exists for 6-64x2-4 and 6-64x4-4");
    h->block_threads = 800;
    h->multi_standalone_noise = name[3] != 0;
  }
  else if (strcmp(name, "multi4u") == 0 || strcmp(name, "multi4u_gen") == 0) {  // ND = 4, six waves (one cost wave)
    if (h->K % 64 != 0) return fail(h, MPPI_ERR_UNSUPPORTED, "multi form needs K to be a multiple of 16 ND");
    if (!h->mfma_ok || !multi_variant_supported(h->hidden, h->n_hidden))
      return fail(h, MPPI_ERR_UNSUPPORTED, "multi form exists for 6-32x2-4, 6-32x4-4 and 6-64x2-4");
    h->block_threads = 1040;
    h->multi_standalone_noise = name[7] != 0;
  }
  else if (strncmp(name, "multi", 5) == 0) {
    const int nd = name[5] - '0';
    const bool gen = strcmp(name + 6, "_gen") == 0;
    if ((nd != 1 && nd != 2 && nd != 4) || (name[6] != 0 && !gen)) return fail(h, MPPI_ERR_INVALID, "unknown variant");
    if (h->K % (16 * nd) != 0) return fail(h, MPPI_ERR_UNSUPPORTED, "multi form needs K to be a multiple of 16 ND");
    if (!h->mfma_ok || !multi_variant_supported(h->hidden, h->n_hidden))
      return fail(h, MPPI_ERR_UNSUPPORTED, "multi form exists for 6-32x2-4, 6-32x4-4 and 6-64x2-4");
    h->block_threads = 1000 + nd;
    h->multi_standalone_noise = gen;
  }
  else if (strcmp(name, "fused") == 0 || strcmp(name, "block256") == 0) h->block_threads = 256;
  else if (strcmp(name, "block64") == 0) h->block_threads = 64;
  else return fail(h, MPPI_ERR_INVALID, "unknown variant");
  return MPPI_OK;
}

}  // extern "C"
