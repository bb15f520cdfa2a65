// abi_pack.hip -- register / LDS images of the network weights for every kernel form, and the MRG32k3a jump tables.
#include "abi_internal.hpp"

using namespace mppi;
using namespace mppi_abi;

namespace mppi_abi {

constexpr uint64_t M1 = 4294967087ULL, M2 = 4294944443ULL;

struct Mat3 {
  uint32_t a[9];
};
Mat3 mat_mul(const Mat3 &A, const Mat3 &B, uint64_t m)
{
  Mat3 R;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      uint64_t acc = 0;
      for (int k = 0; k < 3; k++) acc = (acc + (uint64_t)A.a[3 * i + k] * B.a[3 * k + j] % m) % m;
      R.a[3 * i + j] = (uint32_t)acc;
    }
  return R;
}
Mat3 mat_identity()
{
  Mat3 R{{1, 0, 0, 0, 1, 0, 0, 0, 1}};
  return R;
}
Mat3 mat_pow(Mat3 A, uint64_t e, uint64_t m)
{
  Mat3 R = mat_identity();
  while (e) {
    if (e & 1) R = mat_mul(R, A, m);
    A = mat_mul(A, A, m);
    e >>= 1;
  }
  return R;
}
Mat3 base_A1() { return Mat3{{0, 1, 0, 0, 0, 1, (uint32_t)(M1 - 810728ULL), 1403580u, 0}}; }
Mat3 base_A2() { return Mat3{{0, 1, 0, 0, 0, 1, (uint32_t)(M2 - 1370589ULL), 0, 527612u}}; }

int compute_k99(int K)
{
  // smallest k with (double)k >= .99*NUM_ROLLOUTS, mppi_controller.cu:141
  const double thr = .99 * (double)K;
  int k = 0;
  while (k < K && !((double)k >= thr)) k++;
  return k;
}

// A-operand / bias register image for rollout_mfma.hip (see the mapping comment there).
std::vector<float> pack_mfma_weights(const std::vector<float> &theta, int H, int NHID)
{
  const int MT = H / 16, KSH = H / 4;
  const int nA0 = MT * 2, nAH = MT * KSH, nAL = KSH;
  const int nA = nA0 + (NHID - 1) * nAH + nAL;
  const int nBias = NHID * MT * 4 + 4;
  const int nTree = 4 * KSH + 1;  // mfma_net.hpp: MfmaTree -- the output layer for the butterfly form, behind the MFMA image
  std::vector<float> out((size_t)(nA + nBias + nTree) * 64, 0.0f);
  // offsets of W_l / b_l in theta, layers = 6, H x NHID, 4
  std::vector<int> wo, bo, nin, nout;
  int off = 0, prev = kNetIn;
  for (int l = 0; l <= NHID; l++) {
    const int no = (l < NHID) ? H : kNetOut;
    wo.push_back(off);
    bo.push_back(off + no * prev);
    nin.push_back(prev);
    nout.push_back(no);
    off += no * prev + no;
    prev = no;
  }
  for (int lane = 0; lane < 64; lane++) {
    const int row = lane & 15, kk = lane >> 4;  // A operand: A[row][k = kk]
    const int gq = row >> 2, rq = row & 3;
    // layer 0
    for (int m = 0; m < MT; m++)
      for (int s = 0; s < 2; s++) {
        const int n = 16 * m + 4 * rq + gq, kap = 4 * s + kk;
        out[(size_t)(m * 2 + s) * 64 + lane] = (kap < kNetIn) ? theta[wo[0] + n * kNetIn + kap] : 0.0f;
      }
    for (int l = 1; l < NHID; l++) {
      const int aoff = nA0 + (l - 1) * nAH;
      for (int m = 0; m < MT; m++)
        for (int s = 0; s < KSH; s++) {
          const int n = 16 * m + 4 * rq + gq, kap = 4 * s + kk;
          out[(size_t)(aoff + m * KSH + s) * 64 + lane] = theta[wo[l] + n * H + kap];
        }
    }
    {
      const int aoff = nA0 + (NHID - 1) * nAH;
      for (int s = 0; s < KSH; s++) {
        const int o = row & 3, kap = 4 * s + kk;
        out[(size_t)(aoff + s) * 64 + lane] = theta[wo[NHID] + o * H + kap];
      }
    }
    // biases: lane (j, g) register (m, r) holds D row 16m+4g+r = neuron 16m+4r+g
    const int g = lane >> 4;
    for (int l = 0; l < NHID; l++)
      for (int m = 0; m < MT; m++)
        for (int r = 0; r < 4; r++)
          out[(size_t)(nA + l * MT * 4 + m * 4 + r) * 64 + lane] = theta[bo[l] + 16 * m + 4 * r + g];
    for (int r = 0; r < 4; r++) out[(size_t)(nA + NHID * MT * 4 + r) * 64 + lane] = theta[bo[NHID] + r];
    // tree form of the output layer: lane (j, g) multiplies activation 4 s + g into outputs 0..3; its bias is b3[g]
    for (int s = 0; s < KSH; s++)
      for (int o = 0; o < 4; o++) out[(size_t)(nA + nBias + 4 * s + o) * 64 + lane] = theta[wo[NHID] + o * H + 4 * s + g];
    out[(size_t)(nA + nBias + 4 * KSH) * 64 + lane] = theta[bo[NHID] + g];
  }
  return out;
}

// Register image of the row form (rollout_row.hip: row_load): lane p of a rollout owns neurons 2p, 2p+1 of the hidden layers
// and outputs 2(p&1), 2(p&1)+1; entry i of lane p is float4 index i * 16 + p.  Hidden biases times kTanhScale.
std::vector<float> pack_row_weights(const std::vector<float> &theta)
{
  const int H = 32;
  const float *W1 = theta.data(), *B1 = W1 + H * kNetIn, *W2 = B1 + H, *B2 = W2 + H * H, *W3 = B2 + H, *B3 = W3 + kNetOut * H;
  std::vector<float> out((size_t)row_pack_floats(), 0.0f);
  for (int p = 0; p < 16; p++) {
    const int j0 = 2 * p, j1 = 2 * p + 1, o0 = 2 * (p & 1), o1 = o0 + 1;
    auto entry = [&](int i) { return &out[((size_t)i * 16 + p) * 4]; };
    for (int i = 0; i < 3; i++) {
      float *e = entry(i);
      e[0] = W1[j0 * kNetIn + 2 * i]; e[1] = W1[j1 * kNetIn + 2 * i];
      e[2] = W1[j0 * kNetIn + 2 * i + 1]; e[3] = W1[j1 * kNetIn + 2 * i + 1];
    }
    for (int i = 0; i < H / 2; i++) {
      float *e = entry(3 + i), *f = entry(3 + H / 2 + i);
      e[0] = W2[j0 * H + 2 * i]; e[1] = W2[j1 * H + 2 * i]; e[2] = W2[j0 * H + 2 * i + 1]; e[3] = W2[j1 * H + 2 * i + 1];
      f[0] = W3[o0 * H + 2 * i]; f[1] = W3[o1 * H + 2 * i]; f[2] = W3[o0 * H + 2 * i + 1]; f[3] = W3[o1 * H + 2 * i + 1];
    }
    float *b = entry(35), *c = entry(36);
    b[0] = B1[j0] * kTanhScale; b[1] = B1[j1] * kTanhScale; b[2] = B2[j0] * kTanhScale; b[3] = B2[j1] * kTanhScale;
    c[0] = B3[o0]; c[1] = B3[o1];
    // tree form (row_out_tree): this lane's own two activations into the four outputs, output (p >> 2) ^ i at position i
    const int o = p >> 2;
    float *t0 = entry(37), *t1 = entry(38), *t2 = entry(39);
    t0[0] = W3[o * H + j0]; t0[1] = W3[(o ^ 1) * H + j0]; t0[2] = W3[o * H + j1]; t0[3] = W3[(o ^ 1) * H + j1];
    t1[0] = W3[(o ^ 2) * H + j0]; t1[1] = W3[(o ^ 3) * H + j0]; t1[2] = W3[(o ^ 2) * H + j1]; t1[3] = W3[(o ^ 3) * H + j1];
    t2[0] = B3[o];
  }
  return out;
}

// Image of the 64-wide row form (rollout_row64.hip: row64_load + the LDS part): lane g of a 32-lane rollout owns neurons
// 2g, 2g+1 of every hidden layer; register entry i of lane g at float4 index i * 32 + g, then the 64 x 64 layers as
// [layer][k][g] pairs.  Hidden biases times kTanhScale.  Output layer in the order of row64_out_tree: Q = outputs {0, 1},
// P = outputs {2, 3}, inside each the output the lane keeps at the row_ror:8 level (bit 3 of g) first.
std::vector<float> pack_row64_weights(const std::vector<float> &theta, int NHID)
{
  const int H = 64, NB = (NHID + 1) / 2, NE = 3 + NB + 3;
  std::vector<float> out((size_t)row64_pack_floats(NHID), 0.0f);
  std::vector<const float *> Wl(NHID + 1), Bl(NHID + 1);
  {
    const float *p = theta.data();
    int prev = kNetIn;
    for (int l = 0; l <= NHID; l++) {
      const int no = (l < NHID) ? H : kNetOut;
      Wl[l] = p;
      Bl[l] = p + (size_t)no * prev;
      p += (size_t)no * prev + no;
      prev = no;
    }
  }
  for (int g = 0; g < 32; g++) {
    const int j0 = 2 * g, j1 = 2 * g + 1;
    auto entry = [&](int i) { return &out[((size_t)i * 32 + g) * 4]; };
    for (int i = 0; i < 3; i++) {
      float *e = entry(i);
      e[0] = Wl[0][j0 * kNetIn + 2 * i]; e[1] = Wl[0][j1 * kNetIn + 2 * i];
      e[2] = Wl[0][j0 * kNetIn + 2 * i + 1]; e[3] = Wl[0][j1 * kNetIn + 2 * i + 1];
    }
    for (int l = 0; l < NHID; l++) {
      float *e = entry(3 + l / 2) + 2 * (l & 1);
      e[0] = Bl[l][j0] * kTanhScale; e[1] = Bl[l][j1] * kTanhScale;
    }
    const int row = g >> 4, b3 = (g >> 3) & 1;
    const int qa = b3, qb = 1 - b3, pa = 2 + b3, pb = 3 - b3;
    const float *W3 = Wl[NHID];
    float *q = entry(3 + NB), *pp = entry(4 + NB), *c = entry(5 + NB);
    q[0] = W3[qa * H + j0]; q[1] = W3[qb * H + j0]; q[2] = W3[qa * H + j1]; q[3] = W3[qb * H + j1];
    pp[0] = W3[pa * H + j0]; pp[1] = W3[pb * H + j0]; pp[2] = W3[pa * H + j1]; pp[3] = W3[pb * H + j1];
    c[0] = Bl[NHID][2 * row + b3];
  }
  float *lds = out.data() + (size_t)NE * 32 * 4;
  for (int l = 1; l < NHID; l++)
    for (int k = 0; k < H; k++)
      for (int g = 0; g < 32; g++) {
        float *e = lds + (((size_t)(l - 1) * H + k) * 32 + g) * 2;
        e[0] = Wl[l][(2 * g) * H + k];
        e[1] = Wl[l][(2 * g + 1) * H + k];
      }
  return out;
}

// Image of the 4x4x1-MFMA form (rollout_m44.hip): float4 q of lane l at float4 index q * 64 + l.  Lane l holds ITS neuron's
// rows (B operands) of layer 0 and of the hidden layers, its hidden biases times kTanhScale, and -- for the output layer,
// which reduces over the lanes -- the weights of outputs {0,1} and {2,3} for the four activations 4 (l >> 2) + s of its block.
std::vector<float> pack_m44_weights(const std::vector<float> &theta, int NHID)
{
  const int H = 64, qb = 2, qh = 2 + (NHID + 3) / 4, qo = qh + (NHID - 1) * 16, qt = qo + 5;
  std::vector<float> out((size_t)qt * 64 * 4, 0.0f);
  std::vector<const float *> Wl(NHID + 1), Bl(NHID + 1);
  {
    const float *p = theta.data();
    int prev = kNetIn;
    for (int l = 0; l <= NHID; l++) {
      const int no = (l < NHID) ? H : kNetOut;
      Wl[l] = p;
      Bl[l] = p + (size_t)no * prev;
      p += (size_t)no * prev + no;
      prev = no;
    }
  }
  for (int l = 0; l < 64; l++) {
    auto at = [&](int e) -> float & { return out[((size_t)(e >> 2) * 64 + l) * 4 + (e & 3)]; };
    for (int c = 0; c < kNetIn; c++) at(c) = Wl[0][l * kNetIn + c];
    for (int j = 0; j < NHID; j++) at(4 * qb + j) = Bl[j][l] * kTanhScale;
    for (int j = 1; j < NHID; j++)
      for (int k = 0; k < H; k++) at(4 * (qh + 16 * (j - 1)) + k) = Wl[j][l * H + k];
    const int b = l >> 2;
    const float *W3 = Wl[NHID];
    for (int s = 0; s < 4; s++) {
      at(4 * qo + 2 * s) = W3[0 * H + 4 * b + s];
      at(4 * qo + 2 * s + 1) = W3[1 * H + 4 * b + s];
      at(4 * (qo + 2) + 2 * s) = W3[2 * H + 4 * b + s];
      at(4 * (qo + 2) + 2 * s + 1) = W3[3 * H + 4 * b + s];
    }
    at(4 * (qo + 4)) = Bl[NHID][l >> 4];
  }
  return out;
}

int seed_device(mppi_handle *h, uint64_t seed, uint64_t offset)
{
  // base state: L'Ecuyer's default 12345 x 6, scrambled by the seed (DESIGN.md noise spec)
  uint32_t base[6] = {12345u, 12345u, 12345u, 12345u, 12345u, 12345u};
  if (seed != 0) {
    const uint32_t x1 = ((uint32_t)seed) ^ 0x55555555u;
    const uint32_t x2 = (uint32_t)((seed >> 32) ^ 0xAAAAAAAAu);
    base[0] = (uint32_t)((uint64_t)x1 * base[0] % M1);
    base[1] = (uint32_t)((uint64_t)x2 * base[1] % M1);
    base[2] = (uint32_t)((uint64_t)x1 * base[2] % M1);
    base[3] = (uint32_t)((uint64_t)x2 * base[3] % M2);
    base[4] = (uint32_t)((uint64_t)x1 * base[4] % M2);
    base[5] = (uint32_t)((uint64_t)x2 * base[5] % M2);
  }
  int sub_bits = 0;
  while ((1LL << sub_bits) < (long long)h->K) sub_bits++;
  HIPCHK(h, launch_noise_init(h->d_rng[0], h->K, base, h->d_sub, sub_bits, h->d_one, offset, h->stream));
  h->rng_cur = 0;
  h->cfg.seed = seed;
  return MPPI_OK;
}

int upload_rng_tables(mppi_handle *h)
{
  std::vector<uint32_t> sub(32 * 18), one(64 * 18), jump((size_t)h->noise_C * 18);
  Mat3 a1 = base_A1(), a2 = base_A2();
  for (int b = 0; b < 64; b++) {  // A^(2^b)
    memcpy(&one[(size_t)b * 18], a1.a, 36);
    memcpy(&one[(size_t)b * 18 + 9], a2.a, 36);
    a1 = mat_mul(a1, a1, M1);
    a2 = mat_mul(a2, a2, M2);
  }
  for (int b = 64; b < 76; b++) {
    a1 = mat_mul(a1, a1, M1);
    a2 = mat_mul(a2, a2, M2);
  }
  for (int b = 0; b < 32; b++) {  // A^(2^76 * 2^b)
    memcpy(&sub[(size_t)b * 18], a1.a, 36);
    memcpy(&sub[(size_t)b * 18 + 9], a2.a, 36);
    a1 = mat_mul(a1, a1, M1);
    a2 = mat_mul(a2, a2, M2);
  }
  const Mat3 j1 = mat_pow(base_A1(), 2ULL * (uint64_t)h->noise_L, M1);
  const Mat3 j2 = mat_pow(base_A2(), 2ULL * (uint64_t)h->noise_L, M2);
  Mat3 c1 = mat_identity(), c2 = mat_identity();
  for (int c = 0; c < h->noise_C; c++) {  // A^(2 L c)
    memcpy(&jump[(size_t)c * 18], c1.a, 36);
    memcpy(&jump[(size_t)c * 18 + 9], c2.a, 36);
    c1 = mat_mul(j1, c1, M1);
    c2 = mat_mul(j2, c2, M2);
  }
  HIPCHK(h, hipMemcpy(h->d_sub, sub.data(), sub.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->d_one, one.data(), one.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->d_jump, jump.data(), jump.size() * 4, hipMemcpyHostToDevice));
  return MPPI_OK;
}

}  // namespace mppi_abi
