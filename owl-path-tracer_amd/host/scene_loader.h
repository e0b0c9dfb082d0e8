// scene_loader.h -- scene ingestion of the host entry point: settings.json, <scene>.json, <scene>.obj.scene.
// Mirrors path_tracer/src/utils/parser.cpp, utils/mesh_loader.cpp and application.cpp:143-181 (see scene_loader.cpp).
#pragma once
#include <stdint.h>

#include <array>
#include <string>
#include <vector>

namespace host {

constexpr int kMatFloats = 17;

struct TestData { // parser.hpp:12-23
    std::string name, material_name, attribute_name;
    int material_type = 0;
    std::vector<std::array<float, 3>> vec_values;
    std::vector<float> flt_values;
    float step_size = 0.0f;
};

struct Settings { // parser.hpp:25-37
    std::string scene;
    int buffer_size[2] = {0, 0};
    int max_samples = 0, max_path_depth = 0;
    bool environment_use = false, environment_auto = false;
    float environment_color[3] = {0, 0, 0};
    float environment_intensity = 0.0f;
    TestData test;
};

struct Camera { // camera.hpp:6-12
    float look_from[3], look_at[3], look_up[3], vertical_fov;
};

struct Material {
    std::string name;
    std::array<float, kMatFloats> data; // material_data order, device_global.hpp:19-36
    std::string texture_file;           // "<name>-textures/<filename>" or ""
};

struct Mesh { // mesh_loader.hpp:10-16
    std::string name;
    std::vector<float> vertices, normals, texcoords; // 3, 3, 2 floats per element
    std::vector<int32_t> indices;                    // 3 per triangle
};

struct Entity {
    int mesh = -1, material = -1;
};

struct Scene {
    Camera camera;
    std::vector<Material> materials;
    std::vector<Mesh> meshes;
    std::vector<Entity> entities;
};

Settings parse_settings(const std::string& path);                 // parser.cpp:81-117
void parse_scene_json(const std::string& path, Scene* out);       // parser.cpp:19-78
std::vector<Mesh> load_obj(const std::string& path);              // mesh_loader.cpp:85-121 (+ create_mesh :9-83)
Scene load_scene(const std::string& assets_dir, const std::string& scene_name); // application.cpp:143-181
int attribute_index(const std::string& attribute_name);           // application.cpp:329-360; -1 if unknown

} // namespace host
