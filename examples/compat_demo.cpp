// Host program written against the reference-named types of include/lprt_compat.hpp: loads a scene blob (what a
// loader would have produced), drives Renderer<DataT> the way rt/gui/imgui_window.cu does, writes the RGBA32F frame.
//   compat_demo scene.blob W H frames spp f32|f16 out.raw
#include "lprt_compat.hpp"

#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

using namespace lprt_compat;

template <typename T> static void read_vec(std::ifstream &f, std::vector<T> &v, size_t n) {
    v.resize(n);
    f.read(reinterpret_cast<char *>(v.data()), (std::streamsize)(sizeof(T) * n));
}

template <typename DataT> static int run(std::ifstream &f, const uint32_t cnt[6], int W, int H, int frames, int spp, const char *out_path) {
    auto res = std::make_shared<RDResource<DataT>>();
    read_vec(f, res->vbo_data, cnt[0]);
    read_vec(f, res->indices, cnt[1]);
    read_vec(f, res->objects_vao, cnt[2]);
    std::vector<lprt_material> mats;
    read_vec(f, mats, cnt[3]);
    for (const auto &m : mats) {
        MaterialHolder<DataT> h;
        for (int k = 0; k < 3; k++) {
            h.constants.color[k] = m.color[k];
            h.constants.emission[k] = m.emission[k];
        }
        h.constants.metallic = m.metallic;
        h.constants.roughness = m.roughness;
        h.constants.double_sided = m.double_sided != 0;
        res->materials.push_back(h);
    }
    for (const auto &vao : res->objects_vao) // what load_mesh does per primitive (loader.cu:434)
        res->objects_bvh.push_back(std::make_shared<ObjectBVH<DataT>>(vao.ebo_length / 3));
    std::vector<lprt_object> objs;
    std::vector<lprt_aabb> boxes;
    std::vector<lprt_light> lights;
    read_vec(f, objs, cnt[4]);
    read_vec(f, boxes, cnt[4]);
    read_vec(f, lights, cnt[5]);
    lprt_camera cam;
    float w2c[16];
    f.read(reinterpret_cast<char *>(&cam), sizeof(cam));
    f.read(reinterpret_cast<char *>(w2c), sizeof(w2c));
    if (!f) {
        std::cerr << "short scene blob\n";
        return 2;
    }
    auto scene = std::make_shared<RDScene<DataT>>();
    for (size_t i = 0; i < objs.size(); i++) {
        ObjectConstantData o;
        std::memcpy(o.transform_L2W, objs[i].transform_L2W, 64);
        std::memcpy(o.transform_W2L, objs[i].transform_W2L, 64);
        o.materialid = objs[i].materialid;
        o.objectid = objs[i].objectid;
        scene->objects_constants.push_back(o);
        scene->objects_aabb.push_back({{boxes[i].lower[0], boxes[i].lower[1], boxes[i].lower[2]}, {boxes[i].upper[0], boxes[i].upper[1], boxes[i].upper[2]}});
    }
    for (const auto &l : lights) {
        RDLight<DataT> o;
        o.type = (typename RDLight<DataT>::Type)l.type;
        o.maximum_distance = l.maximum_distance;
        for (int k = 0; k < 3; k++) {
            o.direction[k] = l.direction[k];
            o.intensity[k] = l.intensity[k];
            o.position[k] = l.position[k];
        }
        scene->lights.push_back(o);
    }
    std::memcpy(scene->camera.transform, cam.transform, 64);
    std::memcpy(scene->camera.transform_L2W, cam.transform_L2W, 64);
    scene->camera.field_of_view_y = cam.field_of_view_y;
    scene->camera.aspect_ratio = cam.aspect_ratio;
    scene->camera.z_near = cam.z_near;
    scene->camera.z_far = cam.z_far;

    Renderer<DataT> renderer(W, H, (int)res->objects_vao.size());
    renderer.set_rdresource(res);
    renderer.dynamic_settings.gi_on = true;
    renderer.dynamic_settings.spp = (uint32_t)spp;
    for (int k = 0; k < frames; k++)
        renderer.render(scene, w2c);
    std::vector<float> rgba((size_t)W * H * 4);
    renderer.read_output(rgba.data());
    std::ofstream out(out_path, std::ios::binary);
    out.write(reinterpret_cast<const char *>(rgba.data()), (std::streamsize)(rgba.size() * sizeof(float)));
    const auto &b = *res->objects_bvh.back();
    std::printf("rendered %d frame(s) %dx%d; last BLAS: %u triangles, %u nodes, depth %u\n", frames, W, H, b.n_triangles, b.n_nodes, b.depth);
    return out ? 0 : 3;
}

int main(int argc, char **argv) {
    if (argc != 8) {
        std::cerr << "usage: compat_demo scene.blob W H frames spp f32|f16 out.raw\n";
        return 1;
    }
    std::ifstream f(argv[1], std::ios::binary);
    char magic[8];
    uint32_t cnt[6];
    f.read(magic, 8);
    f.read(reinterpret_cast<char *>(cnt), sizeof(cnt));
    if (!f || std::memcmp(magic, "LPRTSCN1", 8) != 0) {
        std::cerr << "not a scene blob\n";
        return 2;
    }
    const int W = std::atoi(argv[2]), H = std::atoi(argv[3]), frames = std::atoi(argv[4]), spp = std::atoi(argv[5]);
    try {
        if (std::string(argv[6]) == "f16")
            return run<float16>(f, cnt, W, H, frames, spp, argv[7]);
        return run<float>(f, cnt, W, H, frames, spp, argv[7]);
    } catch (const std::exception &e) {
        std::cerr << "error: " << e.what() << "\n";
        return 4;
    }
}
