// Runs the TensorFlow shim's ops (integration/tf_shim/tfgx_tf_ops.cc, linked against the mock runtime of this directory
// and libtfgx.so / libtfgx_dist.so) from a small script — the pipelines integration/tf_shim/tfgx_tf.py composes for a
// tfg.layers.GCN call, spelled by tests/test_gpu_tf_shim.py.  One statement per line:
//
//   input  <name> <float|int32|int64> <host|device> <d0,d1,...|scalar> <file.bin>     raw little-endian file -> tensor
//   comm   <name>                                   a 1-rank ncclComm_t (tfgx_dist_unique_id + tfgx_dist_comm_init) as the
//                                                   host int64 scalar the sharded ops take
//   op     <OpName> in=<a,b,...> out=<x,y,...> [attr=<name>:<int|bool|float>:<value> ...]
//   expect_error <substring> op <OpName> ...        the op must FAIL with a status containing the substring
//   save   <name> <file.bin>                        tensor -> raw file (a one-line shape/dtype header goes to stdout)
//
// usage: tf_shim_mock_driver <script> <workdir>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include "mock_runtime.h"
#include "tfgx_dist.h"

using namespace tensorflow;

static std::vector<std::string> Split(const std::string& s, char sep)
{
    std::vector<std::string> v;
    std::string cur;
    std::istringstream in(s);
    while (std::getline(in, cur, sep)) v.push_back(cur);
    return v;
}

static void Fail(int line, const std::string& m)
{
    std::fprintf(stderr, "mock driver, line %d: %s\n", line, m.c_str());
    std::exit(1);
}

int main(int argc, char** argv)
{
    if (argc < 3) {
        std::fprintf(stderr, "usage: %s <script> <workdir>\n", argv[0]);
        return 2;
    }
    const std::string dir = argv[2];
    hipStream_t stream = nullptr;
    if (hipStreamCreate(&stream) != hipSuccess) Fail(0, "hipStreamCreate failed");
    std::map<std::string, Tensor> env;
    std::ifstream script(argv[1]);
    std::string line;
    int ln = 0, ops_run = 0;
    while (std::getline(script, line)) {
        ++ln;
        std::istringstream in(line);
        std::string cmd;
        if (!(in >> cmd) || cmd[0] == '#') continue;
        std::string want_error;
        if (cmd == "expect_error") {
            in >> want_error >> cmd;
            if (cmd != "op") Fail(ln, "expect_error must be followed by an op statement");
        }
        if (cmd == "input") {
            std::string name, type, place, dims, file;
            in >> name >> type >> place >> dims >> file;
            const DataType dt = type == "float" ? DT_FLOAT : type == "int32" ? DT_INT32 : type == "int64" ? DT_INT64 : DT_INVALID;
            if (dt == DT_INVALID) Fail(ln, "bad dtype " + type);
            std::vector<std::int64_t> d;
            if (dims != "scalar") for (auto& x : Split(dims, ',')) d.push_back(std::atoll(x.c_str()));
            Tensor t(dt, TensorShape(d), place == "host");
            const std::size_t bytes = static_cast<std::size_t>(t.NumElements()) * mock::SizeOf(dt);
            std::vector<char> buf(bytes ? bytes : 1);
            std::ifstream f(dir + "/" + file, std::ios::binary);
            if (bytes && !f.read(buf.data(), static_cast<std::streamsize>(bytes))) Fail(ln, "short read of " + file);
            if (place == "host") std::memcpy(t.raw(), buf.data(), bytes);
            else if (bytes && hipMemcpy(t.raw(), buf.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) Fail(ln, "hipMemcpy H2D");
            env[name] = t;
        } else if (cmd == "comm") {
            std::string name;
            in >> name;
            unsigned char uid[TFGX_DIST_UNIQUE_ID_BYTES];
            void* comm = nullptr;
            if (tfgx_dist_unique_id(uid) != 0 || tfgx_dist_comm_init(1, 0, uid, &comm) != 0) Fail(ln, tfgx_dist_last_error());
            Tensor t(DT_INT64, TensorShape(), true);
            t.flat<std::int64_t>().data()[0] = reinterpret_cast<std::int64_t>(comm);
            env[name] = t;
        } else if (cmd == "op") {
            std::string op, tok;
            in >> op;
            std::vector<std::string> ins, outs;
            AttrMap attrs;
            while (in >> tok) {
                if (tok.rfind("in=", 0) == 0) ins = Split(tok.substr(3), ',');
                else if (tok.rfind("out=", 0) == 0) outs = Split(tok.substr(4), ',');
                else if (tok.rfind("attr=", 0) == 0) {
                    auto p = Split(tok.substr(5), ':');
                    if (p.size() != 3) Fail(ln, "attr=<name>:<type>:<value>");
                    AttrValue v;
                    if (p[1] == "int") { v.kind = AttrValue::kInt; v.i = std::atoll(p[2].c_str()); }
                    else if (p[1] == "bool") { v.kind = AttrValue::kBool; v.b = p[2] == "true"; }
                    else if (p[1] == "float") { v.kind = AttrValue::kFloat; v.f = static_cast<float>(std::atof(p[2].c_str())); }
                    else Fail(ln, "bad attr type " + p[1]);
                    attrs[p[0]] = v;
                } else Fail(ln, "bad token " + tok);
            }
            std::vector<Tensor> inputs, outputs;
            for (auto& n : ins) {
                if (env.find(n) == env.end()) Fail(ln, "no tensor named " + n);
                inputs.push_back(env[n]);
            }
            const Status st = mock::RunOp(op, inputs, attrs, stream, &outputs);
            if (!want_error.empty()) {
                if (st.ok()) Fail(ln, op + " succeeded but an error containing '" + want_error + "' was expected");
                if (st.message().find(want_error) == std::string::npos) Fail(ln, "error was: " + st.message());
                std::printf("expected_error %s: %s\n", op.c_str(), st.message().c_str());
                continue;
            }
            if (!st.ok()) Fail(ln, op + ": " + st.message());
            if (outputs.size() != outs.size()) Fail(ln, op + ": output count differs from out=");
            for (std::size_t i = 0; i < outs.size(); ++i) env[outs[i]] = outputs[i];
            ++ops_run;
        } else if (cmd == "save") {
            std::string name, file;
            in >> name >> file;
            if (env.find(name) == env.end()) Fail(ln, "no tensor named " + name);
            if (hipStreamSynchronize(stream) != hipSuccess) Fail(ln, "hipStreamSynchronize");
            const Tensor& t = env[name];
            const std::size_t bytes = static_cast<std::size_t>(t.NumElements()) * mock::SizeOf(t.dtype());
            std::vector<char> buf(bytes ? bytes : 1);
            if (t.in_host_memory()) std::memcpy(buf.data(), t.raw(), bytes);
            else if (bytes && hipMemcpy(buf.data(), t.raw(), bytes, hipMemcpyDeviceToHost) != hipSuccess) Fail(ln, "hipMemcpy D2H");
            std::ofstream f(dir + "/" + file, std::ios::binary);
            f.write(buf.data(), static_cast<std::streamsize>(bytes));
            std::printf("saved %s dtype=%d dims=", name.c_str(), static_cast<int>(t.dtype()));
            for (int i = 0; i < t.dims(); ++i) std::printf("%s%lld", i ? "," : "", static_cast<long long>(t.dim_size(i)));
            std::printf("\n");
        } else {
            Fail(ln, "unknown statement " + cmd);
        }
    }
    if (hipStreamSynchronize(stream) != hipSuccess) Fail(ln, "hipStreamSynchronize at exit");
    std::printf("TF_SHIM_MOCK_OK ops_run=%d registered=%zu\n", ops_run, mock::RegisteredOps().size());
    return 0;
}
