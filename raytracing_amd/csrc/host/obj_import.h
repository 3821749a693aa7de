// obj_import.h -- OBJImportHelper: mirror of the reference's minimal OBJ reader
// (OBJImportHelper.cs:10-85), plus a binary ".mesh" form of an already-parsed model so that
// scenes can be built where the reference's asset files are not present (the GPU box).
//
// Behaviour kept from OBJImportHelper.cs:19-75:
//   * only "v " and "f" lines are used ("vn"/"vt"/"g"/"#" fall through, line 29/51);
//   * single-space dialect ("v x y z"): vertex = Position + (x, z, y) * scale  (y/z swapped, line 39);
//     double-space dialect ("v  x  y  z"): vertex = Position + (x, y, z) * scale (line 48);
//   * faces are 1-based vertex triples (lines 55-67);
//   * the loop reads a line and tests EndOfStream BEFORE processing it (lines 24-25, 71), so the
//     LAST line of the file is never processed: teapot.KarelModel (1024 f-lines, no trailing
//     newline) yields 1023 triangles.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "scene_elements.h"

namespace rth {

// A parsed model: what ImportModel saw, before scale/position are applied.
struct ParsedModel {
    int dialect = 0;             // 0 = single space (y/z swapped on import), 1 = double space
    std::vector<float> v;        // 3 floats per processed "v " line, as parsed (x, y, z)
    std::vector<int32_t> f;      // 3 one-based indices per processed "f" line
};

class OBJImportHelper {
  public:
    static std::string& AssetDirectory() {
        static std::string dir = ".";
        return dir;
    }
    // OBJImportHelper.cs:81-84 builds "../../../assets/<name>.KarelModel"; here the directory is
    // configurable and a pre-parsed "<name>.mesh" is used when the text asset is absent.
    static std::string FilePath(const std::string& modelName) {
        const std::string text = AssetDirectory() + "/" + modelName + ".KarelModel";
        if (std::ifstream(text).good()) return text;
        return AssetDirectory() + "/" + modelName + ".mesh";
    }

    // vertexTransform (not in the reference): applied to every vertex after scale/position; used by
    // the builder-defined TeapotTable scene to tilt the table (see scenes.h)
    static std::vector<Triangle> ImportModel(const std::string& filePath, float scale, Vector3 Position,
                                             const Material& material,
                                             const std::function<Vector3(Vector3)>& vertexTransform = nullptr) {
        const ParsedModel m = EndsWith(filePath, ".mesh") ? ReadMesh(filePath) : ParseText(filePath);
        std::vector<Vector3> verticies;
        verticies.reserve(m.v.size() / 3);
        for (size_t i = 0; i + 2 < m.v.size(); i += 3) {
            const float x = m.v[i], y = m.v[i + 1], z = m.v[i + 2];
            const Vector3 vtx = Position + (m.dialect == 0 ? Vector3(x, z, y) : Vector3(x, y, z)) * scale;
            verticies.push_back(vertexTransform ? vertexTransform(vtx) : vtx);
        }
        std::vector<Triangle> triangles;
        triangles.reserve(m.f.size() / 3);
        for (size_t i = 0; i + 2 < m.f.size(); i += 3)
            triangles.emplace_back(verticies.at(m.f[i] - 1), verticies.at(m.f[i + 1] - 1), verticies.at(m.f[i + 2] - 1),
                                   material);
        return triangles;
    }

    // StreamReader.ReadLine: terminators \n, \r, \r\n; an unterminated tail is a line
    static std::vector<std::string> ReadLines(const std::string& text) {
        std::vector<std::string> lines;
        size_t i = 0;
        const size_t n = text.size();
        while (i < n) {
            size_t j = text.find_first_of("\r\n", i);
            if (j == std::string::npos) j = n;
            lines.push_back(text.substr(i, j - i));
            if (j < n && text[j] == '\r' && j + 1 < n && text[j + 1] == '\n') j++;
            i = j + 1;
        }
        return lines;
    }

    static ParsedModel ParseText(const std::string& filePath) {
        std::ifstream in(filePath, std::ios::binary);
        if (!in) throw std::runtime_error("FileNotFoundException: " + filePath);
        std::stringstream ss;
        ss << in.rdbuf();
        const std::vector<std::string> lines = ReadLines(ss.str());
        ParsedModel m;
        int dialect = -1;
        auto note = [&](int d) {
            if (dialect >= 0 && dialect != d) throw std::runtime_error("mixed spacing dialects in " + filePath);
            dialect = d;
        };
        for (size_t li = 0; li + 1 < lines.size(); li++) {  // the last line read is never processed
            const std::string& line = lines[li];
            if (line.size() < 3) continue;  // empty lines are skipped (line 27)
            const bool isV = line[0] == 'v' && line[1] == ' ';
            const bool isF = !isV && line[0] == 'f';
            if (!isV && !isF) continue;
            const bool single = line[2] != ' ';
            const std::vector<std::string> tok =
                single ? Split(line.substr(2), " ") : Split(line.substr(3), "  ");
            if (tok.size() < 3) throw std::runtime_error("IndexOutOfRangeException: short line in " + filePath);
            if (isV) {
                note(single ? 0 : 1);
                for (int k = 0; k < 3; k++) m.v.push_back(strtof(tok[k].c_str(), nullptr));  // float.Parse
            } else {
                for (int k = 0; k < 3; k++) {
                    const long idx = strtol(tok[k].c_str(), nullptr, 10);  // int.Parse
                    if (idx < 1 || (size_t)idx > m.v.size() / 3)
                        throw std::runtime_error("ArgumentOutOfRangeException: face index in " + filePath);
                    m.f.push_back((int32_t)idx);
                }
            }
        }
        m.dialect = dialect < 0 ? 0 : dialect;
        return m;
    }

    // ".mesh": "RTMESH1\0", int32 dialect, int32 n_vertices, int32 n_faces, float32 v[3n], int32 f[3m]
    static void WriteMesh(const ParsedModel& m, const std::string& path) {
        std::ofstream out(path, std::ios::binary);
        if (!out) throw std::runtime_error("cannot write " + path);
        const char magic[8] = {'R', 'T', 'M', 'E', 'S', 'H', '1', 0};
        const int32_t hdr[3] = {m.dialect, (int32_t)(m.v.size() / 3), (int32_t)(m.f.size() / 3)};
        out.write(magic, 8);
        out.write((const char*)hdr, sizeof hdr);
        out.write((const char*)m.v.data(), (std::streamsize)(m.v.size() * 4));
        out.write((const char*)m.f.data(), (std::streamsize)(m.f.size() * 4));
    }
    static ParsedModel ReadMesh(const std::string& path) {
        std::ifstream in(path, std::ios::binary);
        if (!in) throw std::runtime_error("FileNotFoundException: " + path);
        char magic[8];
        int32_t hdr[3];
        in.read(magic, 8);
        in.read((char*)hdr, sizeof hdr);
        if (!in || std::memcmp(magic, "RTMESH1", 8) != 0 || hdr[1] < 0 || hdr[2] < 0)
            throw std::runtime_error("not a mesh file: " + path);
        ParsedModel m;
        m.dialect = hdr[0];
        m.v.resize((size_t)hdr[1] * 3);
        m.f.resize((size_t)hdr[2] * 3);
        in.read((char*)m.v.data(), (std::streamsize)(m.v.size() * 4));
        in.read((char*)m.f.data(), (std::streamsize)(m.f.size() * 4));
        if (!in) throw std::runtime_error("truncated mesh file: " + path);
        for (int32_t idx : m.f)
            if (idx < 1 || idx > hdr[1]) throw std::runtime_error("bad face index in " + path);
        return m;
    }

  private:
    static bool EndsWith(const std::string& s, const std::string& suffix) {
        return s.size() >= suffix.size() && s.compare(s.size() - suffix.size(), suffix.size(), suffix) == 0;
    }
    static std::vector<std::string> Split(const std::string& s, const std::string& sep) {
        std::vector<std::string> out;
        size_t i = 0;
        for (;;) {
            size_t j = s.find(sep, i);
            if (j == std::string::npos) {
                out.push_back(s.substr(i));
                return out;
            }
            out.push_back(s.substr(i, j - i));
            i = j + sep.size();
        }
    }
};

}  // namespace rth
