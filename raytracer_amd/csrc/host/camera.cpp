#include "camera.h"

#include <cmath>

namespace rtamd {
namespace host {
namespace {

inline float dot3(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vec3 cross3(Vec3 a, Vec3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
// cgmath InnerSpace::normalize: self * (1 / magnitude)
inline Vec3 normalize3(Vec3 v) {
    float inv = 1.0f / std::sqrt(dot3(v, v));
    return v * inv;
}

}  // namespace

Mat4 Mat4::look_at_rh(Vec3 eye, Vec3 center, Vec3 up) {
    // cgmath 0.18 Matrix4::look_to_rh(eye, center - eye, up)
    Vec3 f = normalize3(center - eye);
    Vec3 s = normalize3(cross3(f, up));
    Vec3 u = cross3(s, f);
    Mat4 r;
    r.m[0] = s.x;  r.m[1] = u.x;  r.m[2] = -f.x;  r.m[3] = 0.0f;
    r.m[4] = s.y;  r.m[5] = u.y;  r.m[6] = -f.y;  r.m[7] = 0.0f;
    r.m[8] = s.z;  r.m[9] = u.z;  r.m[10] = -f.z; r.m[11] = 0.0f;
    r.m[12] = -dot3(eye, s); r.m[13] = -dot3(eye, u); r.m[14] = dot3(eye, f); r.m[15] = 1.0f;
    return r;
}

Mat4 Mat4::perspective(float fovy_rad, float aspect, float near_, float far_) {
    // cgmath 0.18 PerspectiveFov -> Matrix4
    float f = 1.0f / std::tan(fovy_rad / 2.0f);
    Mat4 r{};
    r.m[0] = f / aspect;
    r.m[5] = f;
    r.m[10] = (far_ + near_) / (near_ - far_);
    r.m[11] = -1.0f;
    r.m[14] = (2.0f * far_ * near_) / (near_ - far_);
    return r;
}

Mat4 Mat4::operator*(const Mat4 &r) const {
    Mat4 o;
    for (int c = 0; c < 4; c++)
        for (int row = 0; row < 4; row++) {
            float s = m[0 * 4 + row] * r.m[c * 4 + 0];
            s = s + m[1 * 4 + row] * r.m[c * 4 + 1];
            s = s + m[2 * 4 + row] * r.m[c * 4 + 2];
            s = s + m[3 * 4 + row] * r.m[c * 4 + 3];
            o.m[c * 4 + row] = s;
        }
    return o;
}

bool Mat4::invert(Mat4 *out) const {
    // cofactor expansion (adjugate / determinant)
    const float *a = m;
    float inv[16];
    inv[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] + a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10];
    inv[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] - a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10];
    inv[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] + a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9];
    inv[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] - a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9];
    inv[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] - a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10];
    inv[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] + a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10];
    inv[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] - a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9];
    inv[13] = a[0] * a[9] * a[14] - a[0] * a[10] * a[13] - a[8] * a[1] * a[14] + a[8] * a[2] * a[13] + a[12] * a[1] * a[10] - a[12] * a[2] * a[9];
    inv[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] + a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6];
    inv[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] - a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6];
    inv[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] + a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5];
    inv[14] = -a[0] * a[5] * a[14] + a[0] * a[6] * a[13] + a[4] * a[1] * a[14] - a[4] * a[2] * a[13] - a[12] * a[1] * a[6] + a[12] * a[2] * a[5];
    inv[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] - a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6];
    inv[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] + a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6];
    inv[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] - a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5];
    inv[15] = a[0] * a[5] * a[10] - a[0] * a[6] * a[9] - a[4] * a[1] * a[10] + a[4] * a[2] * a[9] + a[8] * a[1] * a[6] - a[8] * a[2] * a[5];
    float det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12];
    if (det == 0.0f) return false;
    float inv_det = 1.0f / det;
    for (int i = 0; i < 16; i++) out->m[i] = inv[i] * inv_det;
    return true;
}

Camera Camera::new_default(float aspect) {
    const float deg2rad = (float)(3.14159265358979323846 / 180.0);
    return Camera(aspect, 120.0f * deg2rad, Vec3{0, 0, 0}, Vec3{0, 1, 0}, Vec3{1, 0, 0});
}

void Camera::add_position(float x, float y, float z) { position_ = position_ + Vec3{x, y, z}; }

void Camera::set_rotation(float yaw, float pitch) {
    const float pi = 3.14159265358979323846f;
    pitch = f32_max(f32_min(pitch, pi / 2.1f), -pi / 2.1f);
    float sp = std::sin(pitch), cp = std::cos(pitch);
    float sy = std::sin(yaw), cy = std::cos(yaw);
    direction_ = normalize3(Vec3{cy * cp, sp, -sy * cp});
}

CameraViewVectors Camera::generate_view_vectors() const {
    float sin_fov = std::sin(fov_ / 2.0f);
    float cos_fov = std::cos(fov_ / 2.0f);
    Vec3 front = normalize3(direction_);
    Vec3 right = normalize3(cross3(direction_, up_));
    Vec3 up = normalize3(cross3(right, direction_));
    return {right * sin_fov, up * (sin_fov / aspect_), front * cos_fov, position_};
}

CameraViewProjMatrices Camera::view_proj_matrices() const {
    return {Mat4::look_at_rh(position_, position_ + direction_, up_), Mat4::perspective(fov_, aspect_, NEAR, FAR), NEAR, FAR};
}

bool Camera::inv_proj_view(Mat4 *out) const {
    CameraViewProjMatrices vp = view_proj_matrices();
    return (vp.proj * vp.view).invert(out);
}

void generate_primary_rays(const Mat4 &ipv, float near_, float far_, uint32_t width, uint32_t height, float *dir_out) {
    const float *m = ipv.m;
    for (uint32_t py = 0; py < height; py++) {
        float y = ((float)py + 0.5f) / (float)height * 2.0f - 1.0f;
        for (uint32_t px = 0; px < width; px++) {
            float x = ((float)px + 0.5f) / (float)width * 2.0f - 1.0f;
            // v = vec4(x,y,1,1)*far - vec4(x,y,-1,1)*near
            float v0 = x * far_ - x * near_;
            float v1 = y * far_ - y * near_;
            float v2 = 1.0f * far_ - (-1.0f) * near_;
            float v3 = 1.0f * far_ - 1.0f * near_;
            // mat4 * vec4: ((col0*v.x + col1*v.y) + col2*v.z) + col3*v.w
            float d[3];
            for (int r = 0; r < 3; r++) d[r] = ((m[0 + r] * v0 + m[4 + r] * v1) + m[8 + r] * v2) + m[12 + r] * v3;
            // normalize as llvmpipe evaluates it: v * (1 / sqrt(dot)), dot summed z,y,x
            float len2 = (d[2] * d[2] + d[1] * d[1]) + d[0] * d[0];
            float inv = 1.0f / std::sqrt(len2);
            float *o = dir_out + 3 * ((size_t)py * width + px);
            o[0] = d[0] * inv; o[1] = d[1] * inv; o[2] = d[2] * inv;
        }
    }
}

}  // namespace host
}  // namespace rtamd
