// scene_loader.cpp -- see scene_loader.h.  Own JSON reader (json_min.h) and OBJ reader: nlohmann/json and tinyobjloader
// are empty submodules in the reference.  Quirks reproduced on purpose are marked with the reference line.
#include "scene_loader.h"

#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>

#include "json_min.h"

namespace host {

namespace {

std::string slurp(const std::string& path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}

const char* const kFloatFields[14] = {"subsurface", "metallic", "specular", "specular_tint", "roughness", "anisotropic", "sheen", "sheen_tint",
                                      "clearcoat", "clearcoat_gloss", "ior", "specular_transmission", "specular_transmission_roughness", "emission"};
// material_data{} defaults, device_global.hpp:21-35
const float kMatDefault[kMatFloats] = {0.8f, 0.8f, 0.8f, 0.0f, 0.0f, 0.5f, 1.0f, 0.5f, 0.0f, 0.0f, 1.0f, 0.0f, 0.03f, 1.45f, 0.0f, 0.0f, 0.0f};

// tinyobj index triple: v, v/vt, v//vn, v/vt/vn; 1-based, negative = relative to the current count; missing = -1
struct Idx { int v, t, n; };
int fix_index(const char* s, const char* e, int count)
{
    if (s == e) return -1;
    int i = std::atoi(std::string(s, e).c_str());
    return i > 0 ? i - 1 : count + i;
}
Idx parse_idx(const std::string& tok, int nv, int nt, int nn)
{
    const char* s = tok.c_str();
    const char* e = s + tok.size();
    const char* s1 = (const char*)std::memchr(s, '/', (size_t)(e - s));
    if (!s1) return {fix_index(s, e, nv), -1, -1};
    const char* s2 = (const char*)std::memchr(s1 + 1, '/', (size_t)(e - s1 - 1));
    if (!s2) return {fix_index(s, s1, nv), fix_index(s1 + 1, e, nt), -1};
    return {fix_index(s, s1, nv), fix_index(s1 + 1, s2, nt), fix_index(s2 + 1, e, nn)};
}

// mesh_loader.cpp:9-83
Mesh create_mesh(const std::string& name, const std::vector<std::vector<Idx>>& faces, const std::vector<float>& V, const std::vector<float>& VT,
                 const std::vector<float>& VN)
{
    Mesh m;
    m.name = name;
    std::vector<Idx> flat;
    for (auto const& f : faces) flat.insert(flat.end(), f.begin(), f.end());
    std::map<int, int> vertex_mapping; // keyed on the POSITION index only (:44-52)
    size_t off = 0;
    for (size_t f = 0; f < faces.size(); ++f) {
        int tri[3] = {0, 0, 0};
        for (int k = 0; k < 3; ++k) { // only 3 indices per face; the offset advances by 3 whatever the face size (:38,:81)
            if (off + k >= flat.size()) throw std::runtime_error("obj: face index out of range in '" + name + "'");
            Idx ix = flat[off + k];
            if (ix.v < 0 || (size_t)ix.v * 3 + 2 >= V.size()) throw std::runtime_error("obj: vertex index out of range in '" + name + "'");
            auto it = vertex_mapping.find(ix.v);
            if (it == vertex_mapping.end()) {
                it = vertex_mapping.insert({ix.v, (int)(m.vertices.size() / 3)}).first;
                m.vertices.insert(m.vertices.end(), V.begin() + (size_t)ix.v * 3, V.begin() + (size_t)ix.v * 3 + 3);
            }
            tri[k] = it->second;
            if (ix.n >= 0) { // "first seen" normal per local vertex, back-filling earlier attribute-less vertices (:55-66)
                if ((size_t)ix.n * 3 + 2 >= VN.size()) throw std::runtime_error("obj: normal index out of range in '" + name + "'");
                while (m.normals.size() < m.vertices.size()) m.normals.insert(m.normals.end(), VN.begin() + (size_t)ix.n * 3, VN.begin() + (size_t)ix.n * 3 + 3);
            }
            if (ix.t >= 0) { // (:68-78)
                if ((size_t)ix.t * 2 + 1 >= VT.size()) throw std::runtime_error("obj: texcoord index out of range in '" + name + "'");
                while (m.texcoords.size() / 2 < m.vertices.size() / 3) m.texcoords.insert(m.texcoords.end(), VT.begin() + (size_t)ix.t * 2, VT.begin() + (size_t)ix.t * 2 + 2);
            }
        }
        m.indices.insert(m.indices.end(), tri, tri + 3);
        off += 3;
    }
    return m;
}

} // namespace

Settings parse_settings(const std::string& path)
{
    jsonmin::Value c = jsonmin::parse(slurp(path));
    Settings s;
    auto const& t = c.at("test");
    s.test.name = t.at("name").as_string();
    s.test.material_name = t.at("material_name").as_string();
    s.test.attribute_name = t.at("attribute_name").as_string();
    s.test.material_type = t.at("material_type").as_int();
    s.test.step_size = t.at("step_size").as_float();
    for (auto const& v : t.at("values").arr) {
        if (v.is_array()) s.test.vec_values.push_back({v.at(0).as_float(), v.at(1).as_float(), v.at(2).as_float()});
        else s.test.flt_values.push_back(v.as_float());
    }
    s.scene = c.at("scene").as_string();
    s.buffer_size[0] = c.at("buffer_size").at(0).as_int();
    s.buffer_size[1] = c.at("buffer_size").at(1).as_int();
    s.max_path_depth = c.at("max_path_depth").as_int();
    s.max_samples = c.at("max_samples").as_int();
    s.environment_use = c.at("environment_use").as_bool();
    s.environment_auto = c.at("environment_auto").as_bool();
    for (int i = 0; i < 3; ++i) s.environment_color[i] = c.at("environment_color").at(i).as_float();
    s.environment_intensity = c.at("environment_intensity").as_float();
    return s;
}

void parse_scene_json(const std::string& path, Scene* out)
{
    jsonmin::Value c = jsonmin::parse(slurp(path));
    auto const& cam = c.at("camera");
    for (int i = 0; i < 3; ++i) {
        out->camera.look_from[i] = cam.at("look_from").at(i).as_float();
        out->camera.look_at[i] = cam.at("look_at").at(i).as_float();
        out->camera.look_up[i] = cam.at("look_up").at(i).as_float();
    }
    out->camera.vertical_fov = cam.at("vertical_fov").as_float();
    out->materials.clear();
    for (auto const& m : c.at("materials").arr) {
        Material mat;
        mat.name = m.at("name").as_string();
        std::memcpy(mat.data.data(), kMatDefault, sizeof(kMatDefault));
        if (m.at("use_texture").as_bool()) { // parser.cpp:32-35 (a missing key throws, as nlohmann does)
            mat.texture_file = mat.name + "-textures/" + m.at("filename").as_string();
        } else {
            for (int i = 0; i < 3; ++i) mat.data[i] = m.at("base_color").at(i).as_float();
        }
        for (int i = 0; i < 14; ++i) mat.data[3 + i] = m.at(kFloatFields[i]).as_float();
        out->materials.push_back(mat);
    }
}

std::vector<Mesh> load_obj(const std::string& path)
{
    std::ifstream f(path);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::vector<float> V, VT, VN;
    std::vector<Mesh> meshes;
    std::string cur_name;
    std::vector<std::vector<Idx>> cur_faces;
    auto flush = [&]() {
        if (!cur_faces.empty()) meshes.push_back(create_mesh(cur_name, cur_faces, V, VT, VN));
        cur_faces.clear();
    };
    std::string line;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ss(line);
        std::string k;
        ss >> k;
        if (k == "v") {
            float x = 0, y = 0, z = 0;
            ss >> x >> y >> z;
            V.push_back(x); V.push_back(y); V.push_back(z);
        } else if (k == "vn") {
            float x = 0, y = 0, z = 0;
            ss >> x >> y >> z;
            VN.push_back(x); VN.push_back(y); VN.push_back(z);
        } else if (k == "vt") {
            float u = 0, v = 0;
            ss >> u >> v;
            VT.push_back(u); VT.push_back(v);
        } else if (k == "f") {
            std::vector<Idx> face;
            std::string tok;
            while (ss >> tok) face.push_back(parse_idx(tok, (int)(V.size() / 3), (int)(VT.size() / 2), (int)(VN.size() / 3)));
            cur_faces.push_back(face);
        } else if (k == "o" || k == "g") { // tinyobj starts a new shape on both
            flush();
            std::string rest;
            std::getline(ss, rest);
            size_t a = rest.find_first_not_of(" \t");
            cur_name = a == std::string::npos ? "" : rest.substr(a);
            while (!cur_name.empty() && (cur_name.back() == ' ' || cur_name.back() == '\t')) cur_name.pop_back();
        }
    }
    flush();
    return meshes;
}

Scene load_scene(const std::string& assets_dir, const std::string& scene_name)
{
    Scene s;
    parse_scene_json(assets_dir + "/" + scene_name + ".json", &s);
    s.meshes = load_obj(assets_dir + "/" + scene_name + ".obj.scene");
    // application.cpp:166-179: one entity per mesh whose name equals a material name (first match); others vanish
    for (size_t mi = 0; mi < s.meshes.size(); ++mi)
        for (size_t k = 0; k < s.materials.size(); ++k)
            if (s.materials[k].name == s.meshes[mi].name) {
                s.entities.push_back({(int)mi, (int)k});
                break;
            }
    return s;
}

int attribute_index(const std::string& a)
{
    for (int i = 0; i < 13; ++i) // "emission" is not sweepable in the reference (application.cpp:329-360)
        if (a == kFloatFields[i]) return 3 + i;
    return -1;
}

} // namespace host
