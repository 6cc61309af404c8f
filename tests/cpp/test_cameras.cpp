// CPU check of include/sadvio_cameras.hpp: for every camera model of the reference, getRayCamera and project are mutual
// inverses on a pixel grid (project(depth * ray(pixel)) == pixel), the rays are unit vectors, and the validity tests fire.
#include <cstdio>
#include <initializer_list>

#include "sadvio_cameras.hpp"

using namespace sadvio;

int main() {
    int fails = 0;
    auto check = [&](bool ok, const char* what) { if (!ok) { std::printf("FAIL %s\n", what); fails++; } };
    CameraIntrinsics cams[6];
    const char* names[6] = {"pinhole", "equidistant", "equisolid", "stereographic", "omni", "double-sphere"};
    for (auto& c : cams) { c.cx = 376.0; c.cy = 240.0; c.width = 752; c.height = 480; }
    cams[0].kind = CameraKind::Pinhole; cams[0].fx = 458.654; cams[0].fy = 457.296;
    for (int k = 1; k <= 3; k++) { cams[k].fx = cams[k].fy = 1.0; cams[k].rmax = 300.0; }   // angle law with f = 1, pixel scale rmax
    cams[1].kind = CameraKind::FisheyeEquidistant; cams[2].kind = CameraKind::FisheyeEquisolid; cams[3].kind = CameraKind::FisheyeStereographic;
    cams[4].kind = CameraKind::Omni; cams[4].fx = 900.0; cams[4].fy = 905.0; cams[4].xi = 1.7; cams[4].alpha = 1.7 / 2.7;
    cams[5].kind = CameraKind::DoubleSphere; cams[5].fx = 350.0; cams[5].fy = 352.0; cams[5].xi = -0.2; cams[5].alpha = 0.58;
    for (int k = 0; k < 6; k++) {
        double worst = 0.0, worst_norm = 0.0;
        int n = 0;
        for (double u = 40; u < 740; u += 35.5)
            for (double v = 30; v < 470; v += 27.25) {
                double ray[3], pu, pv;
                check(ray_camera(cams[k], u, v, ray), "ray_camera returns true");
                const double nn = std::sqrt(ray[0] * ray[0] + ray[1] * ray[1] + ray[2] * ray[2]);
                worst_norm = std::fmax(worst_norm, std::fabs(nn - 1.0));
                for (double depth : {0.7, 3.0, 12.0}) {
                    const double p[3] = {depth * ray[0], depth * ray[1], depth * ray[2]};
                    if (p[2] < 0.1) continue;   // the projection overload itself refuses those
                    const bool ok = project_camera(cams[k], p, pu, pv);
                    check(ok, "projection of a visible point is valid");
                    worst = std::fmax(worst, std::fmax(std::fabs(pu - u), std::fabs(pv - v)));
                    n++;
                }
            }
        std::printf("%-14s %4d round trips, worst pixel error %.2e, |ray| - 1 <= %.1e\n", names[k], n, worst, worst_norm);
        check(n > 500 && worst < 1e-8 && worst_norm < 1e-12, names[k]);
        double pu, pv;
        const double behind[3] = {0.1, 0.1, -1.0}, far_out[3] = {50.0, 0.0, 1.0};
        check(!project_camera(cams[k], behind, pu, pv), "a point behind the camera is refused");
        if (k == 0) check(!project_camera(cams[k], far_out, pu, pv), "a pinhole point outside the image is refused");
    }
    // Omni with distortion: the reference's inverse model (Heikkila) is approximate: round trip to a fraction of a pixel only
    {
        CameraIntrinsics c = cams[4];
        c.distortion = true; c.D[0] = -0.05; c.D[1] = 0.01; c.D[2] = 1e-4; c.D[3] = -2e-4;
        double worst = 0.0;
        for (double u = 200; u < 560; u += 45)
            for (double v = 120; v < 360; v += 40) {
                double ray[3], pu, pv;
                ray_camera(c, u, v, ray);
                const double p[3] = {4 * ray[0], 4 * ray[1], 4 * ray[2]};
                check(project_camera(c, p, pu, pv), "distorted omni projection valid");
                worst = std::fmax(worst, std::fmax(std::fabs(pu - u), std::fabs(pv - v)));
            }
        std::printf("omni+distortion worst round-trip error %.3f px (first-order inverse distortion)\n", worst);
        check(worst < 0.5, "omni with distortion");
    }
    std::printf("%s\n", fails ? "FAILED" : "PASSED");
    return fails ? 1 : 0;
}
