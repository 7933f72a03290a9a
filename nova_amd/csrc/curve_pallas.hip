// curve_pallas.hip -- the pallas instantiation of the MSM pipeline (one TU per curve keeps hipcc parallel).
#include "curve_impl.hpp"
namespace nmx {
const CurveOps& curve_ops_pallas() {
  static const CurveOps o = CurveImpl<2>::ops();
  return o;
}
}  // namespace nmx
