// Explicit instantiation of the BN254 kernels and host drivers (second field-modulus instantiation of
// the same templates: BASELINE.json configs[3]).
#include "api_impl.cuh"
namespace ark355 {
template struct Api<BnCurve>;
}
