// abi_forms.hip -- which rollout kernel form serves a handle: the selection TABLE (model shape x rollout groups per CU ->
// form, every row with the measurement that justifies it), the forms' properties, their names, mppi_set_rollout_variant.
// tests/test_form_selection_gpu.py times the candidates of every bucket and fails if the table's choice is more than
// 10 % slower than the best of them.
#include "abi_internal.hpp"

using namespace mppi;
using namespace mppi_abi;

namespace mppi_abi {

bool use_mfma(const mppi_handle *h)
{
  if (h->basis || h->pref == Pref::Valu || h->pref == Pref::ValuLds) return false;
  return h->mfma_ok;
}

// "valu" on a standard shape runs the register/scalar-operand kernel; "valu_lds" forces the generic one
bool use_valu_reg(const mppi_handle *h) { return !h->basis && !use_mfma(h) && h->valu_reg_ok && h->pref != Pref::ValuLds; }

namespace {

bool form_supported(Form f, int hidden, int n_hidden)
{
  switch (f) {
    case Form::M44: case Form::M44Chain: return m44_variant_supported(hidden, n_hidden);
    case Form::Row64R16: return row64_variant_supported(hidden, n_hidden);
    case Form::Oct: return oct_variant_supported(hidden, n_hidden);
    case Form::Row: case Form::RowTree: return row_variant_supported(hidden, n_hidden);
    case Form::Multi2: case Form::Multi4: case Form::Multi4Tree:
      return multi_variant_supported(hidden, n_hidden);
    case Form::Quad: case Form::Fused64: case Form::Fused256: return mfma_variant_supported(hidden, n_hidden);
    default: return false;
  }
}

// The network model on a 6 -> hidden x n_hidden -> 4 shape: first row that matches the shape (0 = any), serves the
// handle's 16-rollout groups per CU (0 = any number), exists for the shape and -- when "mfma" asked for the reference's
// summation order (the A/B arm of SURVEY cfg 4) -- keeps that order.  MI355X: 256 CUs of 4 SIMDs; times are rollout-kernel
// times at T = 100 unless a row says otherwise.
struct FormRule {
  int hidden, n_hidden, max_groups_per_cu;
  Form form;
  bool exact;  // the reference's k-ascending sums in every layer: bit-identical to every other exact form
  const char *evidence;
};
const FormRule kFormRules[] = {
    // 64-wide nets in the latency regime: v_mfma_f32_4x4x1 with A-broadcast, every hidden weight in registers, no hand-over
    // between waves (rollout_m44.hip); output layer as a butterfly and -- since the end of round 4 -- the hidden layers as two
    // accumulation chains (K=1920 6-64x4-4 148.5 -> 119.0 us, K=4096 6-64-64-4 70.3 -> 60.6 us, profiles/r04_x_m44_split_ab.txt),
    // inside the north-star tolerance (tests/test_m44_gpu.py); "m44_chain" keeps the reference's order in the hidden layers
    {64, 2, 2, Form::M44, false, "profiles/r04_d_m44_first.txt: K=4096 70.5 us (oct 101, row64 107); K=8192 128 us (oct 166): 110 VGPRs, two workgroups per CU"},
    {64, 4, 2, Form::M44, false, "profiles/r04_d_m44_first.txt: 6-64x4-4 K=1920 149 us (oct 181, row64 227); K=4096 150 us (oct 182); 240 VGPRs = one workgroup per CU, so K=8192 runs in two rounds and still leads: T=60 182 us (oct 217), profiles/r04_e_form_selection.txt"},
    // ... in the reference's order: one M tile per dynamics wave, four of them + four riders (rollout_oct.hip)
    {64, 0, 2, Form::Oct, true, "K=4096: 6-64-64-4 oct 108 us, quad 134; 6-64x4-4 oct 185, quad 279; K=8192: oct 172, multi2 187; 6-64x4-4 oct 365, fused 503; at four groups per CU it loses, 724 vs 508 (profiles/r03_i_*)"},
    // 6-32-32-4 at one group per CU: the recurrence on the vector ALU (rollout_row.hip), output layer as a butterfly
    {32, 2, 2, Form::RowTree, false, "profiles/r04_a_headline_row_tree_*: K=4096 36.7 us (row_exact 45.8, quad 68.0); two groups per CU (106 VGPRs): K=8192 60.3 us (multi2 78.0, multi4 84.2), profiles/r04_e_form_selection.txt; tests/test_row_tree_gpu.py, profiles/r04_b_fuzz_sweep_row_tree_10639_draws.txt"},
    // up to one group per CU: the network split over two SIMDs + a cost and a control wave (rollout_mfma.hip, quad form)
    {0, 0, 1, Form::Quad, true, "6-32-32-4 K=4096: quad 71 us, multi1 / multi2 83, single-wave 122 (profiles/r02_*); row_exact 45.8 is not bit-for-bit needed when \"mfma\" is asked for"},
    // up to two: two dynamics waves (whole network each) + cost + control wave, every wave on a SIMD of its own
    {0, 0, 2, Form::Multi2, true, "K=8192: multi2 77-83 us, quad 112, single-wave 123, row at two groups per CU 93; 6-64-64-4 T=150: 277 vs 359 / 339 us"},
    // beyond: four dynamics waves per workgroup, one per SIMD, riders on the side, eps from the stand-alone generator kernel;
    // the output layer -- 8 of 28 / 16 of 88 matrix instructions per step for 4 useful rows of a 16-row tile -- as a butterfly
    // over the four lanes of a rollout (mfma_net.hpp: nn_last_tree; inside the north-star tolerance, tests/test_multi_tree_gpu.py)
    {0, 0, 0, Form::Multi4Tree, false, "profiles/r04_l_multi4_tree.txt"},
    {0, 0, 0, Form::Multi4, true, "K=16384: 87.5 us (profiles/r03_i_k16384_*) vs 124 single-wave; 6-64-64-4 T=150: 271 us (profiles/r03_d_cfg4_*) vs 341; the in-kernel generator would load one SIMD too much: 341 us"},
    // shapes the multi form does not have (6-64x4-4 beyond two groups per CU): one wave per 16 rollouts does everything, in
    // workgroups of FOUR waves -- the dispatcher spreads a workgroup's waves over the four SIMDs of a CU, whereas 64-thread
    // workgroups are placed one by one and sometimes two on one SIMD (tools/placement_probe.hip: rollout 601 vs 341 us)
    {0, 0, 0, Form::Fused256, true, "6-64x4-4 K=16384: 508 us vs oct 724"},
};

}  // namespace

Form form_of(const mppi_handle *h)
{
  if (h->basis) {
    // basis-function model, wavefronts per 64 rollouts: dynamics + cost + control wave (in-kernel generator) while each gets
    // a SIMD of its own; dynamics + cost wave up to twice that; else one wave
    if (h->forced == Form::Bf1 || h->forced == Form::Bf2 || h->forced == Form::Bf3) return h->forced;
    if (3 * (h->K / 64) <= h->num_simds) return Form::Bf3;
    return (2 * (h->K / 64) <= 2 * h->num_simds) ? Form::Bf2 : Form::Bf1;
  }
  if (!use_mfma(h)) return use_valu_reg(h) ? Form::ValuReg : Form::ValuLds;
  if (h->forced != Form::Auto) return h->forced;
  const int groups = h->K / kRolloutsPerWave, cus = h->num_simds / 4;
  for (const FormRule &r : kFormRules) {
    if ((r.hidden != 0 && r.hidden != h->hidden) || (r.n_hidden != 0 && r.n_hidden != h->n_hidden)) continue;
    if (r.max_groups_per_cu != 0 && groups > r.max_groups_per_cu * cus) continue;
    if (!form_supported(r.form, h->hidden, h->n_hidden)) continue;
    if (h->pref == Pref::Mfma && !r.exact) continue;
    return r.form;
  }
  return Form::Fused256;
}

// oct / multi forms: eps from the stand-alone generator kernel (forced by "_gen", and the automatic choice for ND = 4)
bool form_generator_noise(const mppi_handle *h)
{
  if (h->forced != Form::Auto) return h->multi_standalone_noise;
  const Form f = form_of(h);
  return f == Form::Multi4 || f == Form::Multi4Tree;
}

// does the rollout kernel draw eps itself (a control / noise wavefront with the in-kernel MRG32k3a)?
bool has_noise_wave(const mppi_handle *h)
{
  switch (form_of(h)) {
    case Form::Bf3: case Form::Quad: case Form::Row: case Form::RowTree: case Form::Row64R16: case Form::M44: case Form::M44Chain:
      return true;
    case Form::Oct: case Form::Multi2: case Form::Multi4: case Form::Multi4Tree:
      return !form_generator_noise(h);
    default:
      return false;
  }
}

int form_bf_waves(Form f) { return f == Form::Bf3 ? 3 : f == Form::Bf2 ? 2 : 1; }
int form_multi_nd(Form f)  // launch_rollout_multi's code: 44 = multi4 with the tree output layer
{
  return f == Form::Multi2 ? 2 : f == Form::Multi4 ? 4 : 44;
}
int form_fused_threads(Form f) { return f == Form::Quad ? 512 : f == Form::Fused256 ? 256 : 64; }

}  // namespace mppi_abi

extern "C" {

const char *mppi_rollout_variant(const mppi_handle *h)
{
  if (!h) return "";
  static thread_local char buf[64];
  const Form f = form_of(h);
  const char *gen = form_generator_noise(h) ? "_gen" : "";
  switch (f) {
    case Form::Bf3: return "basis_funcs25_valu_3w";
    case Form::Bf2: return "basis_funcs25_valu_2w";
    case Form::Bf1: return "basis_funcs25_valu";
    case Form::ValuReg: return "valu_reg_lds";
    case Form::ValuLds: return "valu_lds";
    case Form::Multi4Tree: snprintf(buf, sizeof(buf), "mfma16x16x4_h%d_l%d_multi4_tree%s", h->hidden, h->n_hidden, gen); break;
    case Form::Multi2: case Form::Multi4:
      snprintf(buf, sizeof(buf), "mfma16x16x4_h%d_l%d_multi%d%s", h->hidden, h->n_hidden, form_multi_nd(f), gen);
      break;
    case Form::Row: snprintf(buf, sizeof(buf), "valu_row8w_h%d_l%d", h->hidden, h->n_hidden); break;
    case Form::RowTree: snprintf(buf, sizeof(buf), "valu_row8w_tree_h%d_l%d", h->hidden, h->n_hidden); break;
    case Form::Row64R16: snprintf(buf, sizeof(buf), "valu_row64_r16_tree_h%d_l%d", h->hidden, h->n_hidden); break;
    case Form::M44: snprintf(buf, sizeof(buf), "mfma4x4x1_h%d_l%d_m44_split_tree", h->hidden, h->n_hidden); break;
    case Form::M44Chain: snprintf(buf, sizeof(buf), "mfma4x4x1_h%d_l%d_m44_tree", h->hidden, h->n_hidden); break;
    case Form::Oct: snprintf(buf, sizeof(buf), "mfma16x16x4_h%d_l%d_oct8w%s", h->hidden, h->n_hidden, gen); break;
    case Form::Quad: snprintf(buf, sizeof(buf), "mfma16x16x4_h%d_l%d_quad4w", h->hidden, h->n_hidden); break;
    case Form::Fused256: snprintf(buf, sizeof(buf), "mfma16x16x4_h%d_l%d_fused_b256", h->hidden, h->n_hidden); break;
    default: snprintf(buf, sizeof(buf), "mfma16x16x4_h%d_l%d_fused_b64", h->hidden, h->n_hidden); break;
  }
  return buf;
}

int mppi_set_rollout_variant(mppi_handle *h, const char *name)
{
  if (!h || !name) return MPPI_ERR_INVALID;
  auto need = [&](bool ok, const char *what) { return ok ? MPPI_OK : fail(h, MPPI_ERR_UNSUPPORTED, what); };
  int rc = MPPI_OK;
  if (strcmp(name, "auto") == 0) {
    h->pref = Pref::Auto;
    h->forced = Form::Auto;
  }
  else if (strcmp(name, "mfma") == 0) {
    if ((rc = need(h->mfma_ok, "MFMA variant needs 6-HxN-4 with H in {32,64}, N in {2,4}"))) return rc;
    h->pref = Pref::Mfma;
    h->forced = Form::Auto;  // the table again, restricted to the forms in the reference's order: a form forced earlier does not stick
  } else if (strcmp(name, "valu") == 0) { h->pref = Pref::Valu; h->forced = Form::Auto; }
  else if (strcmp(name, "valu_lds") == 0) { h->pref = Pref::ValuLds; h->forced = Form::Auto; }
  else if (strcmp(name, "quad") == 0) h->forced = h->basis ? Form::Bf2 : Form::Quad;
  else if (strcmp(name, "bf3") == 0) {
    if ((rc = need(h->basis, "bf3 is a form of the basis-function model"))) return rc;
    h->forced = Form::Bf3;
  }
  else if (strcmp(name, "row") == 0 || strcmp(name, "row_exact") == 0 || strcmp(name, "row_tree") == 0) {
    if ((rc = need(h->mfma_ok && row_variant_supported(h->hidden, h->n_hidden), "row form exists for 6-32x2-4"))) return rc;
    h->forced = strcmp(name, "row_tree") == 0 ? Form::RowTree : Form::Row;
  }
  else if (strcmp(name, "m44") == 0 || strcmp(name, "m44_chain") == 0) {
    if ((rc = need(h->mfma_ok && m44_variant_supported(h->hidden, h->n_hidden), "m44 form exists for 6-64x2-4 and 6-64x4-4"))) return rc;
    h->forced = name[3] == 0 ? Form::M44 : Form::M44Chain;
  }
  else if (strcmp(name, "row64") == 0 || strcmp(name, "row64_r16") == 0) {  // the vector-ALU arm of the 64-wide A/B
    if ((rc = need(h->mfma_ok && row64_variant_supported(h->hidden, h->n_hidden), "row64 form exists for 6-64x2-4 and 6-64x4-4"))) return rc;
    h->forced = Form::Row64R16;
  }
  else if (strcmp(name, "oct") == 0 || strcmp(name, "oct_gen") == 0) {
    if ((rc = need(h->mfma_ok && oct_variant_supported(h->hidden, h->n_hidden), "oct form exists for 6-64x2-4 and 6-64x4-4"))) return rc;
    h->forced = Form::Oct;
    h->multi_standalone_noise = name[3] != 0;
  }
  else if (strcmp(name, "multi4_tree") == 0 || strcmp(name, "multi4_tree_gen") == 0) {  // ND = 4, butterfly output layer
    if ((rc = need(h->K % 64 == 0, "multi form needs K to be a multiple of 16 ND"))) return rc;
    if ((rc = need(h->mfma_ok && multi_variant_supported(h->hidden, h->n_hidden), "multi form exists for 6-32x2-4, 6-32x4-4 and 6-64x2-4"))) return rc;
    h->forced = Form::Multi4Tree;
    h->multi_standalone_noise = name[11] != 0;
  }
  else if (strncmp(name, "multi", 5) == 0) {
    const int nd = name[5] - '0';
    const bool gen = strcmp(name + 6, "_gen") == 0;
    if ((nd != 2 && nd != 4) || (name[6] != 0 && !gen)) return fail(h, MPPI_ERR_INVALID, "unknown variant");
    if ((rc = need(h->K % (16 * nd) == 0, "multi form needs K to be a multiple of 16 ND"))) return rc;
    if ((rc = need(h->mfma_ok && multi_variant_supported(h->hidden, h->n_hidden), "multi form exists for 6-32x2-4, 6-32x4-4 and 6-64x2-4"))) return rc;
    h->forced = nd == 2 ? Form::Multi2 : Form::Multi4;
    h->multi_standalone_noise = gen;
  }
  else if (strcmp(name, "fused") == 0 || strcmp(name, "block256") == 0) h->forced = h->basis ? Form::Bf1 : Form::Fused256;
  else if (strcmp(name, "block64") == 0) h->forced = h->basis ? Form::Bf1 : Form::Fused64;
  else return fail(h, MPPI_ERR_INVALID, "unknown variant");
  return MPPI_OK;
}

/* Test / tooling hook (not part of the drop-in surface): the rows of the selection table that match this handle's model, as
 * variant names a caller can pass to mppi_set_rollout_variant, best-first; returns how many were written (<= max_n). */
int mppi_debug_form_candidates(const mppi_handle *h, const char **names, int max_n)
{
  if (!h || !names || max_n <= 0) return 0;
  int n = 0;
  auto put = [&](const char *s) { for (int i = 0; i < n; i++) if (strcmp(names[i], s) == 0) return; if (n < max_n) names[n++] = s; };
  if (h->basis) { put("bf3"); put("quad"); put("fused"); return n; }
  if (!h->mfma_ok) { put("valu_lds"); return n; }
  for (const FormRule &r : kFormRules) {
    if ((r.hidden != 0 && r.hidden != h->hidden) || (r.n_hidden != 0 && r.n_hidden != h->n_hidden)) continue;
    if (!form_supported(r.form, h->hidden, h->n_hidden)) continue;
    switch (r.form) {
      case Form::M44: put("m44"); put("m44_chain"); break;
      case Form::Oct: put("oct"); break;
      case Form::RowTree: put("row_tree"); put("row_exact"); break;
      case Form::Quad: put("quad"); break;
      case Form::Multi2: if (h->K % 32 == 0) put("multi2"); break;
      case Form::Multi4Tree: if (h->K % 64 == 0) put("multi4_tree_gen"); break;
      case Form::Multi4: if (h->K % 64 == 0) put("multi4_gen"); break;
      case Form::Fused256: put("fused"); break;
      default: break;
    }
  }
  return n;
}

}  // extern "C"
