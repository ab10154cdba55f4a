// Camera — API kept from src/rendering/camera.rs:9-97 (fov/position/up/direction
// -> view (look_at_rh), proj (perspective, NEAR 0.01, FAR 1000)), minus the window
// handle: the aspect ratio is passed in directly (window.rs supplies width/height).
//
// The matrix helpers restate cgmath 0.18.0 (Cargo.toml:9; source not in the
// reference tree): Matrix4::look_at_rh, perspective(), SquareMatrix::invert.  They
// are plain f32 formulas; camera matrices are INPUTS to the hot path (the same
// matrix / ray arrays are given to the oracle and to the HIP kernels), so no
// parity claim rests on cgmath's internal rounding — "parity unpinned" for the
// last-ulp value of the matrices themselves.
#pragma once
#include "types.h"

namespace rtamd {
namespace host {

constexpr float NEAR = 0.01f;   // camera.rs:6
constexpr float FAR = 1000.0f;  // camera.rs:7

// Column-major 4x4, m[col*4+row], the layout glUniformMatrix4fv(transpose=FALSE)
// receives at src/gl_wrapper/shader.rs:160-162.
struct Mat4 {
    float m[16];
    static Mat4 look_at_rh(Vec3 eye, Vec3 center, Vec3 up);
    static Mat4 perspective(float fovy_rad, float aspect, float near_, float far_);
    Mat4 operator*(const Mat4 &r) const;
    bool invert(Mat4 *out) const;  // false when the determinant is 0 (cgmath returns None)
};

struct CameraViewProjMatrices { Mat4 view, proj; float near_, far_; };  // camera.rs:92-97
struct CameraViewVectors { Vec3 right, up, front, pos; };               // camera.rs:85-90 (deprecated upstream)

class Camera {
public:
    Camera(float aspect, float fov_rad, Vec3 position, Vec3 up, Vec3 direction)
        : aspect_(aspect), fov_(fov_rad), position_(position), up_(up), direction_(direction) {}
    static Camera new_default(float aspect);          // camera.rs:34-42
    void add_position(float x, float y, float z);     // camera.rs:44-46
    void set_rotation(float yaw, float pitch);        // camera.rs:48-56
    void set_position(Vec3 p) { position_ = p; }
    void set_direction(Vec3 d) { direction_ = d; }
    void set_aspect(float a) { aspect_ = a; }
    Vec3 position() const { return position_; }
    Vec3 direction() const { return direction_; }
    CameraViewVectors generate_view_vectors() const;  // camera.rs:58-73
    CameraViewProjMatrices view_proj_matrices() const;  // camera.rs:75-82
    // (proj * view)^-1 as uploaded at src/main.rs:149
    bool inv_proj_view(Mat4 *out) const;

private:
    float aspect_;
    float fov_;
    Vec3 position_, up_, direction_;
};

// Host mirror of res/shaders/ray_trace/ray_create.frag:12 with util/quad-11.vert:8:
// one direction per pixel centre, row 0 = bottom row (GL window convention).
// dir = normalize((invPV * (vec4(x,y,1,1)*far - vec4(x,y,-1,1)*near)).xyz)
void generate_primary_rays(const Mat4 &inv_proj_view, float near_, float far_, uint32_t width, uint32_t height,
                           float *dir_out /* 3*width*height */);

}  // namespace host
}  // namespace rtamd
