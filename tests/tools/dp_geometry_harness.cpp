// TEST INFRASTRUCTURE ONLY: the __host__ __device__ geometry helpers of paramugsy_amd/csrc/dp_internal.hpp (stripes, column groups,
// checkpoint words) compiled for the host, so that tests/test_dp_geometry.py can check their invariants without a GPU.  Never linked
// into the library.
#include "../../paramugsy_amd/csrc/dp_internal.hpp"

using namespace pm;

extern "C" {

long long geo_stripes(long long lb, int C, int tail) { return dp_ck_stripes(lb, C, tail); }
void geo_stripe(long long lb, int C, int tail, long long s, int *jb, int *cs) {
  const DpStripe st = dp_stripe(lb, C, tail, s);
  *jb = st.jb;
  *cs = st.cs;
}
void geo_stripe_of_col(long long lb, int C, int tail, long long j, int *jb, int *cs) {
  const DpStripe st = dp_stripe_of_col(lb, C, tail, j);
  *jb = st.jb;
  *cs = st.cs;
}
long long geo_padded_cols(long long lb, int C, int tail) { return dp_padded_cols(lb, C, tail); }
long long geo_groups(long long lb, int C, int tail) { return dp_ck_groups(lb, C, tail); }
long long geo_words(long long la, long long lb, int C, int tail) { return dp_ck_words(la, lb, C, tail); }
long long geo_bytes_written(long long la, long long lb, int C, int tail) { return dp_ck_bytes_written(la, lb, C, tail); }
long long geo_cost(long long la, long long lb, int C, int tail) { return dp_fill_cost(la, lb, C, tail); }
long long geo_col_word(long long la, long long g, long long t) { return dp_ck_col_word(la, g, t); }
long long geo_row_word(long long la, long long lb, int C, int tail, long long m, long long j) {
  return dp_ck_row_word(la, lb, C, tail, dp_stripe_of_col(lb, C, tail, j), m, j);
}
int geo_group_lane0(long long lb, int C, int tail, long long g) { return dp_group_lane0(C, dp_stripe_of_col(lb, C, tail, g * DP_CK_W * C), g); }
int geo_group_lanes(long long lb, int C, int tail, long long g) { return dp_group_lanes(C, dp_stripe_of_col(lb, C, tail, g * DP_CK_W * C)); }
int geo_ck_r(void) { return DP_CK_R; }
int geo_ck_w(void) { return DP_CK_W; }
long long geo_nck(long long la) { return dp_ck_nck(la); }
}
