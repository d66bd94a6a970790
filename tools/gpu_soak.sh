# Sustained load: 1000 distinct po2-20 segments through the g++-only driver (3 lanes), every seal verified on the host.
#     gpurun -- 'bash tools/gpu_soak.sh <name>'
set -u
O=gpurun_out/${1:-soak}; mkdir -p $O
export TMPDIR=/tmp
python -m zeth_amd.circuits.syn_air syn_a /tmp/syn_a.desc > /dev/null
( time timeout 900 examples/seal_segments --desc /tmp/syn_a.desc --po2 20 --segments 1000 --inflight 3 ) > $O/soak.json 2> $O/soak.err
cat $O/soak.json | cut -c1-400; tail -4 $O/soak.err
rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used\|total" | head -4
