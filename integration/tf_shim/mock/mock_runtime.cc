// Definitions of the mock runtime (see tensorflow/core/framework/op_kernel.h in this directory): registries filled by the
// shim's REGISTER_OP / REGISTER_KERNEL_BUILDER statics, device / host memory for Tensor, and RunOp.  Compiled with hipcc.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <sstream>
#include "mock_runtime.h"

namespace tensorflow {
const char* const DEVICE_GPU = "GPU";
const char* const DEVICE_CPU = "CPU";

namespace mock {
void* DeviceAlloc(std::size_t bytes)
{
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {
        std::fprintf(stderr, "mock runtime: hipMalloc(%zu) failed\n", bytes);
        std::abort();
    }
    return p;
}
void DeviceFree(void* p) { (void)hipFree(p); }
void* HostAlloc(std::size_t bytes) { return std::calloc(1, bytes); }
void HostFree(void* p) { std::free(p); }
std::size_t SizeOf(DataType t)
{
    switch (t) {
        case DT_FLOAT: case DT_INT32: return 4;
        case DT_UINT8: return 1;
        case DT_INT64: return 8;
        default: return 0;
    }
}

namespace {
std::map<std::string, OpDef>& Registry()
{
    static std::map<std::string, OpDef> r;
    return r;
}
std::string Trim(const std::string& s)
{
    const std::size_t a = s.find_first_not_of(" \t"), b = s.find_last_not_of(" \t");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}
DataType TypeOf(const std::string& t)
{
    if (t == "float") return DT_FLOAT;
    if (t == "int32") return DT_INT32;
    if (t == "int64") return DT_INT64;
    if (t == "uint8") return DT_UINT8;
    std::fprintf(stderr, "mock runtime: unknown tensor type '%s'\n", t.c_str());
    std::abort();
}
void SplitSpec(const std::string& spec, std::string* name, std::string* rest)     // "name: rest"
{
    const std::size_t c = spec.find(':');
    *name = Trim(spec.substr(0, c));
    *rest = Trim(spec.substr(c + 1));
}
}  // namespace

const OpDef* FindOp(const std::string& name)
{
    auto it = Registry().find(name);
    return it == Registry().end() ? nullptr : &it->second;
}

std::vector<std::string> RegisteredOps()
{
    std::vector<std::string> v;
    for (auto& kv : Registry()) v.push_back(kv.first);
    return v;
}

Status RunOp(const std::string& name, const std::vector<Tensor>& inputs, const AttrMap& attrs, void* stream,
             std::vector<Tensor>* outputs)
{
    const OpDef* def = FindOp(name);
    if (def == nullptr) return Status("no op named " + name);
    if (!def->make) return Status(name + ": REGISTER_OP without REGISTER_KERNEL_BUILDER");
    if (inputs.size() != def->input_types.size()) {
        std::ostringstream m;
        m << name << ": " << inputs.size() << " inputs given, " << def->input_types.size() << " registered";
        return Status(m.str());
    }
    for (std::size_t i = 0; i < inputs.size(); ++i) {
        if (inputs[i].dtype() != def->input_types[i]) return Status(name + ": input '" + def->input_names[i] + "' has the wrong dtype");
        bool want_host = false;
        for (auto& h : def->host_memory) want_host = want_host || h == def->input_names[i];
        if (inputs[i].in_host_memory() != want_host)
            return Status(name + ": input '" + def->input_names[i] + (want_host ? "' must be in host memory" : "' must be in device memory"));
    }
    AttrMap full = def->attr_defaults;
    for (auto& kv : attrs) {
        auto k = def->attr_kinds.find(kv.first);
        if (k == def->attr_kinds.end()) return Status(name + ": no attr named " + kv.first);
        if (k->second != kv.second.kind) return Status(name + ": attr " + kv.first + " has another type");
        full[kv.first] = kv.second;
    }
    for (auto& kv : def->attr_kinds)
        if (full.find(kv.first) == full.end()) return Status(name + ": attr " + kv.first + " has no default and was not given");
    OpKernelConstruction c(full);
    std::unique_ptr<OpKernel> k(def->make(&c));
    if (!c.status().ok()) return c.status();
    OpKernelContext ctx(inputs, def->output_types, stream);
    k->Compute(&ctx);
    if (!ctx.status().ok()) return ctx.status();
    for (std::size_t i = 0; i < ctx.outputs().size(); ++i)
        if (ctx.outputs()[i].dtype() == DT_INVALID) return Status(name + ": output '" + def->output_names[i] + "' was never allocated");
    *outputs = ctx.outputs();
    return Status();
}
}  // namespace mock

OpRegistrar::OpRegistrar(const OpDefBuilderMock& b)
{
    mock::OpDef& d = mock::Registry()[b.name];
    d.name = b.name;
    std::string n, rest;
    for (auto& s : b.inputs) {
        mock::SplitSpec(s, &n, &rest);
        d.input_names.push_back(n);
        d.input_types.push_back(mock::TypeOf(rest));
    }
    for (auto& s : b.outputs) {
        mock::SplitSpec(s, &n, &rest);
        d.output_names.push_back(n);
        d.output_types.push_back(mock::TypeOf(rest));
    }
    for (auto& s : b.attrs) {                         // "name: int" | "name: int = 0" | "name: bool = true" | "name: float = 1.5"
        mock::SplitSpec(s, &n, &rest);
        std::string type = rest, dflt;
        const std::size_t eq = rest.find('=');
        if (eq != std::string::npos) {
            type = mock::Trim(rest.substr(0, eq));
            dflt = mock::Trim(rest.substr(eq + 1));
        }
        AttrValue v;
        if (type == "int") v.kind = AttrValue::kInt;
        else if (type == "bool") v.kind = AttrValue::kBool;
        else if (type == "float") v.kind = AttrValue::kFloat;
        else {
            std::fprintf(stderr, "mock runtime: attr type '%s' of %s not supported\n", type.c_str(), b.name.c_str());
            std::abort();
        }
        d.attr_kinds[n] = v.kind;
        if (!dflt.empty()) {
            if (v.kind == AttrValue::kInt) v.i = std::atoll(dflt.c_str());
            else if (v.kind == AttrValue::kBool) v.b = dflt == "true";
            else v.f = static_cast<float>(std::atof(dflt.c_str()));
            d.attr_defaults[n] = v;
        }
    }
}

KernelRegistrar::KernelRegistrar(const KernelDefBuilderMock& def, KernelFactory make)
{
    mock::OpDef& d = mock::Registry()[def.name];       // (REGISTER_OP of the same name precedes it in the shim's file)
    d.host_memory = def.host_memory;
    d.make = make;
}
}  // namespace tensorflow
