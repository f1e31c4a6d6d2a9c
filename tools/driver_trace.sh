#!/bin/bash
# HIP API and kernel statistics of i3rcDriver on the step cloud (where does a batch's wall time go?)
B=$GRAFT_REPO_ROOT/i3rc-monte-carlo-model_amd/fortran/build
D=$GRAFT_REPO_ROOT/gpurun_out/drv; mkdir -p $D
$B/makeStepCloudDomain $D/step.dom 32 1.0 > /dev/null
cat > $D/run.nml <<NML
&radiativeTransfer
  solarFlux = 1., solarMu = 1., solarAzimuth = 0., surfaceAlbedo = 0. /
&monteCarlo
  numPhotonsPerBatch = 1000000, numBatches = ${1:-1000}, iseed = 10, nPhaseIntervals = 10001 /
&algorithms
  useRayTracing = .true., useRussianRoulette = .true. /
&output
  reportVolumeAbsorption = .false., reportAbsorptionProfile = .false. /
&fileNames
  domainFileName = "$D/step.dom", outputFluxFile = "$D/flux.txt" /
NML
cd /tmp; export TMPDIR=/tmp
rocprofv3 --hip-runtime-trace --kernel-trace --stats --output-format csv -d $D/trace -- $B/i3rcDriver $D/run.nml > $D/trace.log 2>&1
f=$(ls $D/trace/*/*hip_api_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && column -s, -t < $f | head -20
f=$(ls $D/trace/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && column -s, -t < $f | cut -c1-200 | head -5
