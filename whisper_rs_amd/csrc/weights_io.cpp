// Weight file loading: the reference's `vars.load("weights.ot")` (examples/transcribe.rs:37).  `.ot` is tch's
// named-tensor archive (a zip of pickles) and no sample of it exists in the reference; this build reads the same named
// tensors (VarStore paths = OpenAI state_dict keys) from a safetensors file: 8-byte little-endian
// header length, a JSON header {name: {dtype, shape, data_offsets}}, then the raw little-endian tensor data.
// scripts/convert_checkpoint.py writes such a file from an OpenAI checkpoint.  Tensors the model does not use (e.g.
// OpenAI's encoder.positional_embedding buffer, computed here as in model.rs:395) are skipped, missing ones are
// reported by finalize.
#include "model.h"
#include "json_min.h"
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace wrs {

static float bf16_to_f32(uint16_t v) {
    uint32_t u = (uint32_t)v << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

int load_safetensors(Model& m, const char* path) {
    const int fd = open(path, O_RDONLY);
    WRS_REQUIRE(fd >= 0, "failed to open weights file %s", path);
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size < 8) { close(fd); WRS_THROW(WRS_EINVAL, "weights file %s is too short", path); }
    const size_t size = (size_t)st.st_size;
    void* map = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    WRS_REQUIRE(map != MAP_FAILED, "cannot map weights file %s", path);
    struct Unmap { void* p; size_t n; ~Unmap() { munmap(p, n); } } unmap{map, size};
    const uint8_t* base = (const uint8_t*)map;
    uint64_t hlen = 0;
    memcpy(&hlen, base, 8);
    WRS_REQUIRE(hlen >= 2 && hlen <= size - 8 && hlen < (64u << 20), "weights file %s: bad safetensors header length", path);
    const json::JVal hdr = json::parse((const char*)base + 8, (size_t)hlen);
    WRS_REQUIRE(hdr.kind == json::JVal::Obj, "weights file %s: header is not a JSON object", path);
    const uint8_t* data = base + 8 + hlen;
    const size_t data_size = size - 8 - (size_t)hlen;
    int loaded = 0;
    std::vector<float> tmp;
    for (auto& kv : hdr.o) {
        if (kv.first == "__metadata__") continue;
        std::vector<int64_t> want;
        if (!expected_shape(m, kv.first, want)) continue;  // not a tensor of this architecture
        const json::JVal* dt = kv.second.get("dtype");
        const json::JVal* sh = kv.second.get("shape");
        const json::JVal* off = kv.second.get("data_offsets");
        WRS_REQUIRE(dt && dt->kind == json::JVal::Str && sh && sh->kind == json::JVal::Arr && off && off->kind == json::JVal::Arr && off->a.size() == 2,
                    "weights file: bad entry for %s", kv.first.c_str());
        std::vector<int64_t> shape;
        int64_t n = 1;
        for (auto& d : sh->a) { shape.push_back((int64_t)d.num); n *= (int64_t)d.num; }
        const size_t b = (size_t)off->a[0].num, e = (size_t)off->a[1].num;
        const size_t esz = dt->s == "F32" ? 4 : (dt->s == "F16" || dt->s == "BF16") ? 2 : 0;
        WRS_REQUIRE(esz != 0, "weights file: tensor %s has dtype %s (F32, F16 or BF16 expected)", kv.first.c_str(), dt->s.c_str());
        WRS_REQUIRE(b <= e && e <= data_size && e - b == (size_t)n * esz, "weights file: tensor %s has inconsistent offsets", kv.first.c_str());
        if (dt->s == "BF16") {
            tmp.resize((size_t)n);
            const uint16_t* src = (const uint16_t*)(data + b);
            for (int64_t i = 0; i < n; ++i) tmp[(size_t)i] = bf16_to_f32(src[i]);
            m.set_tensor(kv.first, WRS_F32, shape.data(), (int)shape.size(), tmp.data(), WRS_MEM_HOST);
        } else {
            m.set_tensor(kv.first, dt->s == "F32" ? WRS_F32 : WRS_F16, shape.data(), (int)shape.size(), data + b, WRS_MEM_HOST);
        }
        ++loaded;
    }
    return loaded;
}

}  // namespace wrs
