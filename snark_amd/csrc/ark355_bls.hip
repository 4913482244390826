// Explicit instantiation of the BLS12-381 kernels and host drivers.
#include "api_impl.cuh"
namespace ark355 {
template struct Api<BlsCurve>;
}
