# End-to-end wall time of the Fortran drivers on the GPU box (process start, tables, batches, result files) on the step cloud in the
# reference generator's shape (32 x 1 x 32: one launch of 1e8 photons runs at 2.65e9 photons/s), and the steady-state rate of the
# batch loop from the difference between 1000 and 10 batches.
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
echo "i3rcDriver $nb batches x 1e6 photons: $(( (t1 - t0) / 1000000 )) ms wall"; eval own_$nb=$(( (t1 - t0) / 1000000 ))
if [ -x $B/monteCarloDriver_ref ]; then t0=$(date +%s%N); $B/monteCarloDriver_ref $D/run.nml > $D/outref_$nb.txt 2>&1 || true; t1=$(date +%s%N); echo "reference driver (unchanged) $nb batches: $(( (t1 - t0) / 1000000 )) ms wall"; eval ref_$nb=$(( (t1 - t0) / 1000000 )); fi
done
python3 -c "print('steady state of the batch loop, (1000 - 10 batches): i3rcDriver %.3f ms per batch = %.3e photons/s; unchanged reference driver %.3f ms per batch = %.3e photons/s' % (($own_1000 - $own_10) / 990.0, 990e9 / ($own_1000 - $own_10), (${ref_1000:-0} - ${ref_10:-0}) / 990.0, 990e9 / max(1, ${ref_1000:-0} - ${ref_10:-0})))"
