#!/bin/bash
# Builds libtfgx_tf_ops.so where a TensorFlow-ROCm installation exists (not in this repo's image).
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
python -c "import tensorflow" 2>/dev/null || { echo "TensorFlow is not installed: the shim cannot be built here"; exit 2; }
TF_CFLAGS=$(python -c 'import tensorflow as tf; print(" ".join(tf.sysconfig.get_compile_flags()))')
TF_LFLAGS=$(python -c 'import tensorflow as tf; print(" ".join(tf.sysconfig.get_link_flags()))')
/opt/rocm/bin/hipcc -std=c++17 -shared -fPIC -O2 -DTENSORFLOW_USE_ROCM=1 \
  -I"$ROOT/include" $TF_CFLAGS "$ROOT/integration/tf_shim/tfgx_tf_ops.cc" \
  -L"$ROOT/tf_geometric_amd/lib" -ltfgx -Wl,-rpath,"$ROOT/tf_geometric_amd/lib" $TF_LFLAGS \
  -o "$ROOT/integration/tf_shim/libtfgx_tf_ops.so"
echo "built; load with tf.load_op_library('$ROOT/integration/tf_shim/libtfgx_tf_ops.so')"
