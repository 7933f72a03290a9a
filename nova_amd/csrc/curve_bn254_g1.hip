// curve_bn254_g1.hip -- the bn254_g1 instantiation of the MSM pipeline (one TU per curve keeps hipcc parallel).
#include "curve_impl.hpp"
namespace nmx {
const CurveOps& curve_ops_bn254_g1() {
  static const CurveOps o = CurveImpl<0>::ops();
  return o;
}
}  // namespace nmx
