// sadvio_cameras.hpp — the camera models of SaDVIO on the HOST side of the boundary (header-only C++17).
//
// The reference's analytic pixel factor only exists for the pinhole `Camera` (the `project(…, J_frame, J_lmk)` overloads of
// `Fisheye`, `Omni` and `DoubleSphere` are stubs: fisheye.cpp:352-405, DoubleSphere.cpp:125-133), so its non-pinhole data sets
// run on the model-free ANGULAR factor, whose measurement is the unit bearing `getRayCamera(pixel)`. The device path takes
// bearings as they come (sadvio_flat_window.obs_meas), hence everything model-specific is what this header restates:
//   ray_camera()      ImageSensor::getRayCamera of each model (Camera.cpp:15-25, fisheye.cpp:47-124, DoubleSphere.cpp:14-31)
//   project_camera()  the `project(T_w_lmk, model, scale, p2ds)` overloads that ALandmark::chi2err calls, with each model's own
//                     validity tests (Camera.cpp:26-52, fisheye.cpp:127-172, 195-240, DoubleSphere.cpp:33-78)
// Quirks kept: Fisheye divides pixel offsets by `rmax` while the angle law uses K(0,0) as focal length; its ray is 0/0 at
// the principal point; Omni lifts with (1 - alpha) / f and applies `p + distort(p)` once in this overload.
#pragma once
#include <cmath>

namespace sadvio {

enum class CameraKind { Pinhole, FisheyeEquidistant, FisheyeEquisolid, FisheyeStereographic, Omni, DoubleSphere };

struct CameraIntrinsics {
    CameraKind kind = CameraKind::Pinhole;
    double fx = 1, fy = 1, cx = 0, cy = 0;   // K (ASensor.cpp:17)
    double width = 0, height = 0;            // _raw_data.cols / rows
    double rmax = 1;                         // Fisheye::_rmax (Fisheye.h:48)
    double xi = 0, alpha = 0;                // Omni: alpha = xi / (1 + xi) (Fisheye.h:56); DoubleSphere: both given
    bool distortion = false;                 // Omni only
    double D[4] = {0, 0, 0, 0};              // k1 k2 p1 p2
};

// getRayCamera: unit bearing of pixel (u, v) in the camera frame. Returns false for an unknown model.
inline bool ray_camera(const CameraIntrinsics& c, double u, double v, double ray[3]) {
    auto normalise = [&]() { const double n = std::sqrt(ray[0] * ray[0] + ray[1] * ray[1] + ray[2] * ray[2]); ray[0] /= n; ray[1] /= n; ray[2] /= n; };
    switch (c.kind) {
    case CameraKind::Pinhole:   // Camera.cpp:15-25
        ray[0] = (u - c.cx) / c.fx; ray[1] = (v - c.cy) / c.fy; ray[2] = 1.0;
        normalise();
        return true;
    case CameraKind::FisheyeEquidistant:
    case CameraKind::FisheyeEquisolid:
    case CameraKind::FisheyeStereographic: {   // fisheye.cpp:47-72
        const double xd = (u - c.cx) / c.rmax, yd = (v - c.cy) / c.rmax;
        const double rd = std::sqrt(xd * xd + yd * yd);
        double theta;
        if (c.kind == CameraKind::FisheyeEquidistant) theta = rd / c.fx;
        else if (c.kind == CameraKind::FisheyeEquisolid) theta = 2.0 * std::asin(rd / (2.0 * c.fx));
        else theta = 2.0 * std::atan2(rd, 2.0 * c.fx);
        ray[0] = xd; ray[1] = yd; ray[2] = rd / std::tan(theta);
        normalise();
        return true;
    }
    case CameraKind::Omni: {   // fisheye.cpp:76-124 (inverse distortion after Heikkila, lift after Mei)
        const double mx_d = ((u - c.cx) * (1.0 - c.alpha)) / c.fx, my_d = ((v - c.cy) * (1.0 - c.alpha)) / c.fy;
        double mx = mx_d, my = my_d;
        if (c.distortion) {
            const double k1 = c.D[0], k2 = c.D[1], p1 = c.D[2], p2 = c.D[3];
            const double mx2 = mx_d * mx_d, my2 = my_d * my_d, mxy = mx_d * my_d, rho2 = mx2 + my2, rho4 = rho2 * rho2;
            const double rad = k1 * rho2 + k2 * rho4;
            const double Dx = mx_d * rad + p2 * (rho2 + 2.0 * mx2) + 2.0 * p1 * mxy;
            const double Dy = my_d * rad + p1 * (rho2 + 2.0 * my2) + 2.0 * p2 * mxy;
            const double inv = 1.0 / (1.0 + 4.0 * k1 * rho2 + 6.0 * k2 * rho4 + 8.0 * p1 * my_d + 8.0 * p2 * mx_d);
            mx = mx_d - inv * Dx; my = my_d - inv * Dy;
        }
        const double r2 = mx * mx + my * my;
        if (c.xi == 1.0) {
            const double l = 2.0 / (r2 + 1.0);
            ray[0] = l * mx; ray[1] = l * my; ray[2] = l - 1.0;
        } else {
            const double l = (c.xi + std::sqrt(1.0 + (1.0 - c.xi * c.xi) * r2)) / (1.0 + r2);
            ray[0] = l * mx; ray[1] = l * my; ray[2] = l - c.xi;
        }
        return true;   // on the unit sphere by construction (not re-normalised in the reference either)
    }
    case CameraKind::DoubleSphere: {   // DoubleSphere.cpp:14-31 (Usenko et al.)
        const double mx = (u - c.cx) / c.fx, my = (v - c.cy) / c.fy, r2 = mx * mx + my * my;
        const double mz = (1.0 - c.alpha * c.alpha * r2) / (c.alpha * std::sqrt(1.0 - (2.0 * c.alpha - 1.0) * r2) + 1.0 - c.alpha);
        const double mz2 = mz * mz;
        const double k = (mz * c.xi + std::sqrt(mz2 + (1.0 - c.xi * c.xi) * r2)) / (mz2 + r2);
        ray[0] = k * mx; ray[1] = k * my; ray[2] = k * mz - c.xi;
        return true;
    }
    }
    return false;
}

// project(T_w_lmk, model, scale, p2ds) for a point p already in the CAMERA frame: pixel (u, v) and the model's validity
// verdict (false = ALandmark::chi2err counts the feature 1000).
inline bool project_camera(const CameraIntrinsics& c, const double p[3], double& u, double& v) {
    const double x = p[0], y = p[1], z = p[2];
    auto in_image = [&]() { return !(u < 0 || v < 0 || u > c.width || v > c.height) && std::isfinite(u) && std::isfinite(v); };
    switch (c.kind) {
    case CameraKind::Pinhole:   // Camera.cpp:26-52
        u = (c.fx * x + c.cx * z) / z; v = (c.fy * y + c.cy * z) / z;
        return !(z < 0.1) && in_image();
    case CameraKind::FisheyeEquidistant:
    case CameraKind::FisheyeEquisolid:
    case CameraKind::FisheyeStereographic: {   // fisheye.cpp:127-172
        const double r = std::sqrt(x * x + y * y + z * z), theta = std::acos(z / r), al = std::atan2(y, x);
        double rd;
        if (c.kind == CameraKind::FisheyeEquidistant) rd = c.fx * theta;
        else if (c.kind == CameraKind::FisheyeEquisolid) rd = 2.0 * c.fx * std::sin(theta / 2.0);
        else rd = 2.0 * c.fx * std::tan(theta / 2.0);
        u = rd * std::cos(al) * c.rmax + c.cx; v = rd * std::sin(al) * c.rmax + c.cy;
        return !(z < 0.01) && in_image();
    }
    case CameraKind::Omni: {   // fisheye.cpp:195-240
        if (z < 0.1) { u = v = 0.0; return false; }
        const double d = std::sqrt(x * x + y * y + z * z), zz = z + c.xi * d;
        double px = x / zz, py = y / zz;
        if (c.distortion) {   // Omni::distort returns p + d (fisheye.cpp:176-193)
            const double k1 = c.D[0], k2 = c.D[1], p1 = c.D[2], p2 = c.D[3];
            const double mx2 = px * px, my2 = py * py, mxy = px * py, rho2 = mx2 + my2, rad = k1 * rho2 + k2 * rho2 * rho2;
            const double dx = px * rad + 2.0 * p1 * mxy + p2 * (rho2 + 2.0 * mx2), dy = py * rad + 2.0 * p2 * mxy + p1 * (rho2 + 2.0 * my2);
            px += dx; py += dy;
        }
        u = c.fx * px / (1.0 - c.alpha) + c.cx; v = c.fy * py / (1.0 - c.alpha) + c.cy;
        const double w = c.alpha <= 0.5 ? c.alpha / (1.0 - c.alpha) : (1.0 - c.alpha) / c.alpha;
        return !(z <= -w * d) && in_image();
    }
    case CameraKind::DoubleSphere: {   // DoubleSphere.cpp:33-78
        if (z < 0.1) { u = v = 0.0; return false; }
        const double d1 = std::sqrt(x * x + y * y + z * z), zs = c.xi * d1 + z, d2 = std::sqrt(x * x + y * y + zs * zs);
        const double den = c.alpha * d2 + (1.0 - c.alpha) * zs;
        u = c.fx * (x / den) + c.cx; v = c.fy * (y / den) + c.cy;
        const double w1 = c.alpha <= 0.5 ? c.alpha / (1.0 - c.alpha) : (1.0 - c.alpha) / c.alpha;
        const double w2 = (w1 + c.xi) / std::sqrt(2.0 * w1 * c.xi + c.xi * c.xi + 1.0);
        return !(z <= -w2 * d1) && in_image();
    }
    }
    return false;
}

}  // namespace sadvio
