// C++ name-compatibility layer over the C ABI of lprt.h.
//
// The reference's loader (rt/rtrt/loader.cu), scene manager (rt/rtrt/hierarchy.hpp) and frame driver
// (rt/gui/imgui_window.cu:78-367) talk to the renderer through a handful of host-side types.  This header offers
// types of the same names, members and call order, so that code written against them keeps compiling on a
// Linux + ROCm host with nothing but <lprt.h> behind it — no CUDA, Thrust, cuRAND or OpenGL:
//
//   reference type (file:line)                                   here
//   RDVertex<float>            rt/rtrt/memory.hpp:26-41          layout-identical to lprt_vertex
//   CUDAMeshVAO                rt/rtrt/memory.hpp:560-566        layout-identical to lprt_mesh_vao
//   CUDAMaterial / MaterialHolder  memory.hpp:303-332, 371-380   host constants + optional RDTexture
//   RDTexture(data,w,h,sRGB)   memory.hpp:334-369                keeps the RGBA8 pixels; uploaded on refresh_materials
//   RDLight / RDCamera / ObjectConstantData / RDSkybox / RDScene  memory.hpp:43-51, 279-301, 382-405, 671-716
//   ObjectBVH<DataT>(positions, indices)  rt/trace/object_bvh.hpp:49-63   a ticket; the GPU builds all of them at once
//   RDResource<DataT>          memory.hpp:568-648                vbo_data / indices / objects_vao / objects_bvh / materials
//   CUDARenderGIData<DataT>    rt/rtrt/cuda.hpp:635-792          refresh_materials, fill_objects_cuda_vao, fill_objects_bvh,
//                                                                get_render_input (scene hand-over + TLAS build)
//   RTRTProcedureBuffer<DataT,N>  cuda.hpp:1475-1634             per-pixel state lives in the context; clear_* keep their names
//   DemoSetting                cuda.hpp:2161-2168
//   Renderer<DataT>            rt/gui/imgui_window.cu:78-367     render(rd_scene, world_to_clip)
//
// DataT is `float` or the tag type `float16` (arithmetic in half precision happens on the device; host-side data
// stays float exactly like RDVertex<float> / RDScene<float> in the reference's loader).  Matrices are ROW-major
// float[16] as in the reference's Mat4 (glm users transpose: rt/rtrt/memory.hpp:285-301).
// Errors: every failing lprt_* call throws std::runtime_error(lprt_last_error) — RT_CHECK_CUDA's behaviour
// (rt/utils/cu_exception.cu:9-13).
#pragma once
#include "lprt.h"

#include <array>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

namespace lprt_compat {

struct float16 {}; // DataT tag of the half-precision instantiation

template <typename DataT> struct precision_of;
template <> struct precision_of<float> { static constexpr lprt_precision value = LPRT_F32; };
template <> struct precision_of<float16> { static constexpr lprt_precision value = LPRT_F16; };

inline void check(lprt_ctx *ctx, int rc, const char *what) {
    if (rc != LPRT_OK)
        throw std::runtime_error(std::string(what) + ": " + lprt_last_error(ctx));
}

// One context per renderer instance, shared by the objects below.
struct Context {
    lprt_ctx *ctx = nullptr;
    uint32_t width, height;
    Context(int device, uint32_t w, uint32_t h, uint32_t row_begin, uint32_t row_end, lprt_precision p) : width(w), height(h) {
        check(nullptr, lprt_create(&ctx, device, w, h, row_begin, row_end, p), "lprt_create");
    }
    ~Context() { lprt_destroy(ctx); }
    Context(const Context &) = delete;
    void operator=(const Context &) = delete;
};

template <typename T> using Vec3 = std::array<T, 3>;
template <typename T> using Vec2 = std::array<T, 2>;

template <typename DataT> struct RDVertex { // host vertices are float in both instantiations
    Vec3<float> position, normal, tangent, color;
    Vec2<float> uv0, uv1;
};
static_assert(sizeof(RDVertex<float>) == sizeof(lprt_vertex), "RDVertex must match lprt_vertex");

struct CUDAMeshVAO {
    uint32_t ebo_index, vbo_index, ebo_offset, ebo_length;
};
static_assert(sizeof(CUDAMeshVAO) == sizeof(lprt_mesh_vao), "CUDAMeshVAO must match lprt_mesh_vao");

struct RDTexture {
    std::vector<uint8_t> pixels;
    size_t width, height;
    bool sRGB;
    lprt_ctx *uploaded_to = nullptr;
    uint32_t handle = 0;
    RDTexture(const uint8_t *data, size_t w, size_t h, bool srgb = false) : pixels(data, data + w * h * 4), width(w), height(h), sRGB(srgb) {}
    uint32_t ensure(lprt_ctx *ctx) {
        if (uploaded_to != ctx) {
            check(ctx, lprt_upload_texture(ctx, pixels.data(), (uint32_t)width, (uint32_t)height, sRGB ? 1u : 0u, &handle), "lprt_upload_texture");
            uploaded_to = ctx;
        }
        return handle;
    }
};

template <typename DataT> struct CUDAMaterial {
    Vec3<float> color{1, 1, 1}, emission{0, 0, 0};
    float metallic = 0, roughness = 1;
    bool double_sided = true;
    uint32_t uv_color = UINT32_MAX;
};
template <typename DataT> struct MaterialHolder {
    CUDAMaterial<DataT> constants;
    std::shared_ptr<RDTexture> tex_color;
};

template <typename DataT> struct RDLight {
    enum Type : uint32_t { SPOT = LPRT_LIGHT_SPOT, POINT = LPRT_LIGHT_POINT, DIRECTIONAL = LPRT_LIGHT_DIRECTIONAL } type = POINT;
    float maximum_distance = 1e5f;
    Vec3<float> direction{0, 0, -1}, intensity{1, 1, 1}, position{0, 0, 0};
};
template <typename DataT> struct RDCamera {
    float transform[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};     // world -> view
    float transform_L2W[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}; // view -> world
    float field_of_view_y = 1.0f, aspect_ratio = 1.0f, z_near = 0.1f, z_far = 100.f;
};
struct ObjectConstantData {
    float transform_L2W[16], transform_W2L[16];
    uint32_t materialid = 0, objectid = 0;
};
template <typename DataT> struct RDSkybox {
    std::shared_ptr<RDTexture> texture;
    float delta_x = 0, delta_y = 0, exposure = 1;
};
template <typename DataT> struct RDScene {
    std::vector<ObjectConstantData> objects_constants;
    std::vector<std::pair<Vec3<float>, Vec3<float>>> objects_aabb;
    std::vector<RDLight<DataT>> lights;
    RDCamera<DataT> camera;
    RDSkybox<DataT> skybox;
};

// The reference builds one CPU BVH per primitive in its constructor.  Here construction only records the request;
// CUDARenderGIData::fill_objects_bvh() builds every primitive's BLAS on the GPU in one go and fills in the result.
template <typename DataT> class ObjectBVH {
  public:
    ObjectBVH(const std::vector<Vec3<float>> &positions, const std::vector<uint32_t> &indices) : n_triangles((uint32_t)indices.size() / 3) {
        (void)positions;
    }
    explicit ObjectBVH(uint32_t triangles) : n_triangles(triangles) {}
    uint32_t n_triangles = 0, n_nodes = 0, depth = 0;
    bool built = false;
};

template <typename DataT> struct RDResource {
    std::vector<MaterialHolder<DataT>> materials;
    std::vector<RDVertex<float>> vbo_data; // the one shared VBO load_gltf2 fills (loader.cu:786-802)
    std::vector<uint32_t> indices;         // the one shared EBO, indices already rebased (loader.cu:330-332)
    std::vector<CUDAMeshVAO> objects_vao;
    std::vector<std::shared_ptr<ObjectBVH<DataT>>> objects_bvh;
    // M_shift / M_shift_f32 / position_f32 are derived on the device from the uploaded vertices
    void compute_m(const std::vector<RDVertex<float>> &, const std::vector<uint32_t> &) {}
};

template <typename DataT> struct CUDARenderGIInput { // what the kernels received by value; here: a receipt
    uint32_t n_vaos = 0, n_lights = 0, n_objects = 0;
};

template <typename DataT> struct CUDARenderGIData {
    std::shared_ptr<Context> context;
    std::shared_ptr<RDResource<DataT>> rd_resource;
    std::shared_ptr<RDScene<DataT>> rd_scene;
    float transform_W2C[16] = {};

    CUDARenderGIData() = default;
    CUDARenderGIData(const CUDARenderGIData &) = delete;
    void operator=(const CUDARenderGIData &) = delete;

    void map() {}   // no GL interop behind the boundary
    void unmap() {}

    void refresh_materials() {
        lprt_ctx *c = context->ctx;
        std::vector<lprt_material> m(rd_resource->materials.size());
        for (size_t i = 0; i < m.size(); i++) {
            const auto &h = rd_resource->materials[i];
            std::memset(&m[i], 0, sizeof(lprt_material));
            for (int k = 0; k < 3; k++) {
                m[i].color[k] = h.constants.color[k];
                m[i].emission[k] = h.constants.emission[k];
            }
            m[i].metallic = h.constants.metallic;
            m[i].roughness = h.constants.roughness;
            m[i].double_sided = h.constants.double_sided ? 1u : 0u;
            m[i].uv_color = h.tex_color ? h.constants.uv_color : UINT32_MAX;
            m[i].tex_color = h.tex_color ? h.tex_color->ensure(c) : 0u;
        }
        check(c, lprt_upload_materials(c, m.data(), (uint32_t)m.size()), "lprt_upload_materials");
    }
    void fill_objects_cuda_vao() {
        lprt_ctx *c = context->ctx;
        const auto &r = *rd_resource;
        check(c, lprt_upload_geometry(c, reinterpret_cast<const lprt_vertex *>(r.vbo_data.data()), (uint32_t)r.vbo_data.size(), r.indices.data(),
                                      (uint32_t)r.indices.size(), reinterpret_cast<const lprt_mesh_vao *>(r.objects_vao.data()),
                                      (uint32_t)r.objects_vao.size()),
              "lprt_upload_geometry");
    }
    void fill_objects_bvh() {
        lprt_ctx *c = context->ctx;
        check(c, lprt_build_blas(c), "lprt_build_blas");
        auto &bvhs = rd_resource->objects_bvh;
        bvhs.resize(rd_resource->objects_vao.size());
        for (size_t i = 0; i < bvhs.size(); i++) {
            if (!bvhs[i])
                bvhs[i] = std::make_shared<ObjectBVH<DataT>>(rd_resource->objects_vao[i].ebo_length / 3);
            check(c, lprt_get_blas_info(c, (uint32_t)i, &bvhs[i]->n_triangles, &bvhs[i]->n_nodes, &bvhs[i]->depth), "lprt_get_blas_info");
            bvhs[i]->built = true;
        }
    }
    // per frame: objects + lights + camera + skybox hand-over and the TLAS build (SceneBVH::update)
    CUDARenderGIInput<DataT> get_render_input(bool refresh = true) {
        lprt_ctx *c = context->ctx;
        if (refresh)
            refresh_materials();
        const auto &s = *rd_scene;
        std::vector<lprt_object> obj(s.objects_constants.size());
        std::vector<lprt_aabb> box(obj.size());
        for (size_t i = 0; i < obj.size(); i++) {
            std::memcpy(obj[i].transform_L2W, s.objects_constants[i].transform_L2W, 64);
            std::memcpy(obj[i].transform_W2L, s.objects_constants[i].transform_W2L, 64);
            obj[i].materialid = s.objects_constants[i].materialid;
            obj[i].objectid = s.objects_constants[i].objectid;
            for (int k = 0; k < 3; k++) {
                box[i].lower[k] = s.objects_aabb[i].first[k];
                box[i].upper[k] = s.objects_aabb[i].second[k];
            }
        }
        std::vector<lprt_light> lights(s.lights.size());
        for (size_t i = 0; i < lights.size(); i++) {
            std::memset(&lights[i], 0, sizeof(lprt_light));
            lights[i].type = (uint32_t)s.lights[i].type;
            lights[i].maximum_distance = s.lights[i].maximum_distance;
            for (int k = 0; k < 3; k++) {
                lights[i].direction[k] = s.lights[i].direction[k];
                lights[i].intensity[k] = s.lights[i].intensity[k];
                lights[i].position[k] = s.lights[i].position[k];
            }
        }
        lprt_camera cam;
        std::memset(&cam, 0, sizeof(cam));
        std::memcpy(cam.transform, s.camera.transform, 64);
        std::memcpy(cam.transform_L2W, s.camera.transform_L2W, 64);
        cam.field_of_view_y = s.camera.field_of_view_y;
        cam.aspect_ratio = s.camera.aspect_ratio;
        cam.z_near = s.camera.z_near;
        cam.z_far = s.camera.z_far;
        lprt_skybox sky;
        std::memset(&sky, 0, sizeof(sky));
        if (s.skybox.texture) {
            sky.valid = 1;
            sky.texture = s.skybox.texture->ensure(c);
            sky.delta_x = s.skybox.delta_x;
            sky.delta_y = s.skybox.delta_y;
            sky.exposure = s.skybox.exposure;
        }
        check(c, lprt_set_scene(c, obj.data(), box.data(), (uint32_t)obj.size(), lights.data(), (uint32_t)lights.size(), &cam, transform_W2C,
                                s.skybox.texture ? &sky : nullptr),
              "lprt_set_scene");
        CUDARenderGIInput<DataT> out;
        out.n_vaos = (uint32_t)rd_resource->objects_vao.size();
        out.n_lights = (uint32_t)lights.size();
        out.n_objects = (uint32_t)obj.size();
        return out;
    }
};

// The per-pixel work buffers are the context's; the reference's per-frame clear_* calls (imgui_window.cu:137-141,
// 187) are part of lprt_render_frame.  clear_last_frame_L2W — called when a new model is loaded — drops all history.
template <typename DataT, size_t MAX_DIRECT_LIGHT> struct RTRTProcedureBuffer {
    static_assert(MAX_DIRECT_LIGHT == 4, "the kernels are built for MAX_DIRECT_LIGHT = 4 (rt/gui/imgui_window.cu:76)");
    std::shared_ptr<Context> context;
    int width, height, max_bounces, object_count;
    RTRTProcedureBuffer(int w, int h, int bounces, int objects) : width(w), height(h), max_bounces(bounces), object_count(objects) {}
    RTRTProcedureBuffer(const RTRTProcedureBuffer &) = delete;
    void operator=(const RTRTProcedureBuffer &) = delete;
    void clear_intensity() {}
    void clear_trace_gi_commands() {}
    void clear_di_intensity() {}
    void clear_shade_commands() {}
    void clear_filterable_cache() {}
    void clear_last_frame_L2W() { check(context->ctx, lprt_reset_history(context->ctx), "lprt_reset_history"); }
};

struct DemoSetting {
    bool add_direct_out = true, add_gi_colored = true, add_gi_white = true, demodulate = false, svgf = true;
};

template <typename DataT> class Renderer {
  public:
    static constexpr size_t MAX_DIRECT_LIGHT = 4;
    struct Settings {
        bool gi_on = false;
        bool traced_primary_ray = true; // always: there is no rasteriser behind the boundary
        float svgf_color_mix_weight = 0.1f, svgf_moments_mix_weight = 0.1f, taa_mix_weight = 1.f;
        uint32_t spp = 1;
        const uint64_t *seeds = nullptr; // 3 per sample; null: the library's frame-indexed seeds (the reference calls rand())
    } dynamic_settings;
    DemoSetting demo_setting;
    std::shared_ptr<Context> context;
    std::shared_ptr<CUDARenderGIData<DataT>> render_data;
    std::shared_ptr<RTRTProcedureBuffer<DataT, MAX_DIRECT_LIGHT>> buffer;

    Renderer(int width, int height, int number_of_vaos, int device = 0) : width_(width), height_(height) {
        context = std::make_shared<Context>(device, (uint32_t)width, (uint32_t)height, 0u, (uint32_t)height, precision_of<DataT>::value);
        render_data = std::make_shared<CUDARenderGIData<DataT>>();
        render_data->context = context;
        buffer = std::make_shared<RTRTProcedureBuffer<DataT, MAX_DIRECT_LIGHT>>(width, height, 2, number_of_vaos);
        buffer->context = context;
    }
    // imgui_window.cu:332-337
    void set_rdresource(std::shared_ptr<RDResource<DataT>> res) {
        render_data->rd_resource = std::move(res);
        render_data->refresh_materials();
        render_data->fill_objects_cuda_vao();
        render_data->fill_objects_bvh();
        buffer->clear_last_frame_L2W();
    }
    // imgui_window.cu:339-347: world_to_clip is ROW-major (the GUI passes glm::transpose(proj * view), :293-294)
    void render(std::shared_ptr<RDScene<DataT>> rd_scene, const float world_to_clip[16]) {
        render_data->rd_scene = std::move(rd_scene);
        std::memcpy(render_data->transform_W2C, world_to_clip, 64);
        render_data->map();
        buffer->clear_intensity();
        buffer->clear_di_intensity();
        buffer->clear_shade_commands();
        buffer->clear_trace_gi_commands();
        buffer->clear_filterable_cache();
        render_data->get_render_input(false);
        lprt_frame_settings fs;
        std::memset(&fs, 0, sizeof(fs));
        fs.spp = dynamic_settings.spp;
        fs.gi_on = dynamic_settings.gi_on;
        fs.svgf = demo_setting.svgf;
        fs.add_direct_out = demo_setting.add_direct_out;
        fs.add_gi_colored = demo_setting.add_gi_colored;
        fs.add_gi_white = demo_setting.add_gi_white;
        fs.demodulate = demo_setting.demodulate;
        fs.svgf_color_mix_weight = dynamic_settings.svgf_color_mix_weight;
        fs.svgf_moments_mix_weight = dynamic_settings.svgf_moments_mix_weight;
        fs.taa_mix_weight = dynamic_settings.taa_mix_weight;
        fs.seeds = dynamic_settings.seeds;
        check(context->ctx, lprt_render_frame(context->ctx, &fs), "lprt_render_frame");
        render_data->unmap();
    }
    // tex_draw_target (imgui_window.cu:351, 640): RGBA32F, width*height*4 floats, row 0 at the bottom
    void read_output(float *rgba) {
        check(context->ctx, lprt_read_buffer(context->ctx, LPRT_BUF_OUTPUT, rgba, (size_t)width_ * height_ * 4), "lprt_read_buffer");
    }
    void *output_device_ptr() {
        void *p = nullptr;
        size_t n = 0;
        check(context->ctx, lprt_output_device_ptr(context->ctx, &p, &n), "lprt_output_device_ptr");
        return p;
    }

  private:
    int width_, height_;
};

} // namespace lprt_compat
