// curve_vesta.hip -- the vesta instantiation of the MSM pipeline (one TU per curve keeps hipcc parallel).
#include "curve_impl.hpp"
namespace nmx {
const CurveOps& curve_ops_vesta() {
  static const CurveOps o = CurveImpl<3>::ops();
  return o;
}
}  // namespace nmx
