P="python tools/twin_probe.py --no-kernels --only raster,tiled"
$P
$P --opt recon_pair_streams=2
$P --opt recon_lanes=2
$P --opt recon_pair_streams=2 --opt recon_lanes=2
$P --fuse 15
$P --fuse 30
$P --fuse 6
$P --fuse 0
