#!/usr/bin/env bash
# TEST INFRASTRUCTURE ONLY -- builds oracle/_ref/libref.so, the reference's OWN CPU
# bodies for this path, from the sources where they lie under /root/reference.
#
# The reference's .cpp files as a whole need TensorFlow 1.13 headers (absent here) and
# are unbuildable; but the functions below are TF-free free functions / loops inside
# those files.  They are streamed by line range straight from /root/reference into
# g++'s stdin -- no reference source is written to disk or into this repo -- together
# with system headers and an extern "C" wrapper so ctypes can find the symbols.  The
# only artefact is the shared object under oracle/_ref/ (git-ignored, travels to the
# GPU box with gpurun like our own .so files).
#
#   nnsearch                       tf_ops/CD/tf_nndistance.cpp:21-43
#   NnDistanceGrad CPU loop        tf_ops/CD/tf_nndistance.cpp:126-163 (body of Compute)
#   approxmatch_cpu/matchcost_cpu/matchcostgrad_cpu
#                                  pc_distance/tf_approxmatch.cpp:23-140
#   threenn_cpu/threeinterpolate_cpu/threeinterpolate_grad_cpu
#                                  tf_ops/interpolation/tf_interpolate.cpp:57-153
#   query_ball_point_cpu/group_point_cpu/group_point_grad_cpu
#                                  tf_ops/grouping/query_ball_point.cpp:17-84
#
# Flags follow the reference's own compile scripts (g++ -std=c++11 -O2, no -march, so
# no FMA contraction: tf_ops/CD/tf_cd_compile_abi.sh, pc_distance/makefile); -fopenmp only
# gives meaning to the pragmas of the wrapper loops at the end (the bodies hold none).
set -euo pipefail
REF=${RFNET_REFERENCE:-/root/reference}
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
OUT="$HERE/_ref"
if [ ! -d "$REF" ]; then
    echo "build_ref.sh: $REF not present (GPU box?) -- keeping any prebuilt $OUT/libref.so" >&2
    exit 0
fi
mkdir -p "$OUT"
{
    cat <<'EOF'
#include <algorithm>
#include <vector>
#include <cmath>
#include <math.h>
#include <cstring>
#include <cstdio>
using namespace std;
extern "C" {
EOF
    # nnsearch (declared static in the reference; drop the keyword so it is exported)
    sed -n '21,43p' "$REF/tf_ops/CD/tf_nndistance.cpp" | sed 's/^static void nnsearch/void nnsearch/'
    # the NnDistanceGrad CPU loop is the tail of a Compute() method: give it a signature
    echo 'void ref_nn_distance_grad(int b,int n,int m,const float*xyz1,const float*xyz2,const float*grad_dist1,const int*idx1,const float*grad_dist2,const int*idx2,float*grad_xyz1,float*grad_xyz2){'
    sed -n '126,163p' "$REF/tf_ops/CD/tf_nndistance.cpp"
    echo '}'
    sed -n '23,140p' "$REF/pc_distance/tf_approxmatch.cpp"
    sed -n '57,153p' "$REF/tf_ops/interpolation/tf_interpolate.cpp"
    sed -n '17,84p' "$REF/tf_ops/grouping/query_ball_point.cpp"
    # OpenMP-over-batch wrappers around the UNTOUCHED bodies above ("an OpenMP-over-batch run on
    # all host cores, with core count printed"): each thread calls the reference function on ONE batch
    # element (b = 1) of its own; the reference op itself is single-threaded (tf_nndistance.cpp:79-80).
    cat <<'OMP'
#include <omp.h>
int ref_omp_max_threads(void){ return omp_get_max_threads(); }
void ref_nn_forward_omp(int b,int n,int m,const float*xyz1,const float*xyz2,float*dist1,int*idx1,float*dist2,int*idx2,int threads){
#pragma omp parallel for schedule(dynamic,1) num_threads(threads)
    for (int i=0;i<b;i++){
        nnsearch(1,n,m,xyz1+(size_t)i*n*3,xyz2+(size_t)i*m*3,dist1+(size_t)i*n,idx1+(size_t)i*n);
        nnsearch(1,m,n,xyz2+(size_t)i*m*3,xyz1+(size_t)i*n*3,dist2+(size_t)i*m,idx2+(size_t)i*m);
    }
}
void ref_nn_grad_omp(int b,int n,int m,const float*xyz1,const float*xyz2,const float*gd1,const int*idx1,const float*gd2,const int*idx2,float*g1,float*g2,int threads){
#pragma omp parallel for schedule(dynamic,1) num_threads(threads)
    for (int i=0;i<b;i++)
        ref_nn_distance_grad(1,n,m,xyz1+(size_t)i*n*3,xyz2+(size_t)i*m*3,gd1+(size_t)i*n,idx1+(size_t)i*n,gd2+(size_t)i*m,idx2+(size_t)i*m,g1+(size_t)i*n*3,g2+(size_t)i*m*3);
}
OMP
    echo '}'
} | g++ -x c++ -std=c++11 -O2 -fopenmp -fPIC -shared -o "$OUT/libref.so" -
echo "built $OUT/libref.so"
