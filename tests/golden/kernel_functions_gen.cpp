// Golden-vector generator for the SPH kernel functions: evaluates the REFERENCE's own W3h / GradW3h
// (`__host__ __device__` functions of src/chrono_fsi/sph/physics/SphGeneral.cuh:76-229) as host code, on a fixed grid of inputs.
// The reference header is included where it lies under /root/reference (build container only; nothing of it is copied here);
// make_kernel_golden.sh configures its two generated headers with the reference's own templates and compiles this file.
// Output (stdout, binary float32): for each kernel type t in 0..3, kernel length h in H, sample i in 0..NR-1:
//   r, W3h(t, r, 1/h), and for each direction c in 0..2: d.xyz (= r * DIR[c], rounded to float), GradW3h(t, d, 1/h).xyz
#include <cstdio>
#include <vector>

#include "chrono_fsi/sph/physics/SphGeneral.cuh"

using namespace chrono::fsi::sph;

int main() {
    const float H[3] = {0.01f, 0.012f, 0.025f};
    const float DIR[3][3] = {{1.0f, 0.0f, 0.0f}, {0.6f, -0.48f, 0.64f}, {-0.26726124f, 0.53452248f, 0.80178373f}};
    const int NR = 96;
    std::vector<float> out;
    for (int t = 0; t < 4; t++)
        for (int ih = 0; ih < 3; ih++) {
            const float h = H[ih];
            const float invh = 1.0f / h;
            for (int i = 0; i < NR; i++) {
                const float r = (2.2f * h) * (float)i / (float)(NR - 1);
                out.push_back(r);
                out.push_back((float)W3h((KernelType)t, (Real)r, (Real)invh));
                for (int c = 0; c < 3; c++) {
                    const Real3 d = mR3((Real)(r * DIR[c][0]), (Real)(r * DIR[c][1]), (Real)(r * DIR[c][2]));
                    const Real3 g = GradW3h((KernelType)t, d, (Real)invh);
                    out.push_back((float)d.x); out.push_back((float)d.y); out.push_back((float)d.z);
                    out.push_back((float)g.x); out.push_back((float)g.y); out.push_back((float)g.z);
                }
            }
        }
    fwrite(out.data(), sizeof(float), out.size(), stdout);
    fprintf(stderr, "sizeof(Real) = %zu, %zu floats\n", sizeof(Real), out.size());
    return sizeof(Real) == 4 ? 0 : 1;
}
