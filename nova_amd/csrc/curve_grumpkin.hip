// curve_grumpkin.hip -- the grumpkin instantiation of the MSM pipeline (one TU per curve keeps hipcc parallel).
#include "curve_impl.hpp"
namespace nmx {
const CurveOps& curve_ops_grumpkin() {
  static const CurveOps o = CurveImpl<1>::ops();
  return o;
}
}  // namespace nmx
