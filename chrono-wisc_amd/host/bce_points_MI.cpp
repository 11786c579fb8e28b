// Prints the point set of one BCE generator of the C++ facade as raw doubles (x y z per point): test helper, no GPU needed.
//   bce_points_MI <d0> <layers> box sx sy sz | sphere r polar | cylinder r h polar | annulus ri ro h polar | mesh <file>   > points.bin
//   mesh file: int32 nv, int32 nf, nv x 3 doubles, nf x 3 int32
#include <cstdio>
#include <cstdlib>
#include <string>
#include "ChFsiFluidSystemSPH_MI.h"
using namespace chrono_mi;
using namespace chrono_mi::fsi::sph;
int main(int argc, char* argv[]) {
    if (argc < 5) return 2;
    ChFsiFluidSystemSPH sys;
    sys.SetInitialSpacing(atof(argv[1]));
    sys.SetNumBCELayers(atoi(argv[2]));
    std::string kind = argv[3];
    std::vector<ChVector3d> pts;
    auto a = [&](int i) { return atof(argv[i]); };
    if (kind == "box" && argc >= 7) pts = sys.CreatePointsBoxInterior(ChVector3d(a(4), a(5), a(6)));
    else if (kind == "sphere" && argc >= 6) pts = sys.CreatePointsSphereInterior(a(4), atoi(argv[5]) != 0);
    else if (kind == "cylinder" && argc >= 7) pts = sys.CreatePointsCylinderInterior(a(4), a(5), atoi(argv[6]) != 0);
    else if (kind == "annulus" && argc >= 8) pts = sys.CreatePointsCylinderAnnulus(a(4), a(5), a(6), atoi(argv[7]) != 0);
    else if (kind == "mesh" && argc >= 5) {
        FILE* f = fopen(argv[4], "rb");
        int nv = 0, nf = 0;
        if (!f || fread(&nv, 4, 1, f) != 1 || fread(&nf, 4, 1, f) != 1) return 2;
        std::vector<double> v(3 * (size_t)nv);
        std::vector<int> faces(3 * (size_t)nf);
        if (fread(v.data(), 8, v.size(), f) != v.size() || fread(faces.data(), 4, faces.size(), f) != faces.size()) return 2;
        fclose(f);
        std::vector<ChVector3d> verts(nv);
        for (int i = 0; i < nv; i++) verts[i] = ChVector3d(v[3 * i], v[3 * i + 1], v[3 * i + 2]);
        pts = sys.CreatePointsMesh(verts, faces);
    } else return 2;
    for (const auto& p : pts) fwrite(p.v, sizeof(double), 3, stdout);
    return 0;
}
