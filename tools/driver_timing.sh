# End-to-end wall time of the Fortran drivers on the GPU box (process start, tables, batches, result files).
set -e
B=$GRAFT_REPO_ROOT/i3rc-monte-carlo-model_amd/fortran/build
D=$GRAFT_REPO_ROOT/gpurun_out/drv; mkdir -p $D
$B/makeStepCloudDomain $D/step.dom 32 1.0 > /dev/null
for nb in 10 100 1000; do
cat > $D/run.nml <<NML
&radiativeTransfer
  solarFlux = 1., solarMu = 1., solarAzimuth = 0., surfaceAlbedo = 0. /
&monteCarlo
  numPhotonsPerBatch = 1000000, numBatches = $nb, iseed = 10, nPhaseIntervals = 10001 /
&algorithms
  useRayTracing = .true., useRussianRoulette = .true. /
&output
  reportVolumeAbsorption = .false., reportAbsorptionProfile = .false. /
&fileNames
  domainFileName = "$D/step.dom", outputFluxFile = "$D/flux.txt" /
NML
t0=$(date +%s%N); $B/i3rcDriver $D/run.nml > $D/out_$nb.txt 2>&1 || true; t1=$(date +%s%N)
echo "i3rcDriver $nb batches x 1e6 photons: $(( (t1 - t0) / 1000000 )) ms wall"
if [ -x $B/monteCarloDriver_ref ]; then t0=$(date +%s%N); $B/monteCarloDriver_ref $D/run.nml > $D/outref_$nb.txt 2>&1 || true; t1=$(date +%s%N); echo "reference driver (unchanged) $nb batches: $(( (t1 - t0) / 1000000 )) ms wall"; fi
done
grep -i "cpu time" $D/out_100.txt
