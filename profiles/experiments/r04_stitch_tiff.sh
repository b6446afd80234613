# round 4: `oip stitch` of two 7500 x 25000 x 4 LZW TIFFs (the reference's step 5, imageop.h:365-457) with the device codec
# (default) against the host's threads (OIP_TIFF_GPU_LZW=0); files in tmpfs
D=/dev/shm/oip_stitch_tiff; rm -rf $D; mkdir -p $D
LZW_H=25000 LZW_OUT=$D/a.TIFF ./profiles/experiments/lzw_bench
cp $D/a.TIFF $D/b.TIFF
for m in 1 0 1 0; do
  rm -f $D/out.TIFF
  S=$(date +%s.%N)
  OIP_TIFF_GPU_LZW=$m LOGFILE=$D/oip.log ./opticalimageprocessor_amd/lib/oip stitch --image1 $D/a.TIFF --image2 $D/b.TIFF --fold-cols 50 -o $D/out.TIFF > $D/stdout.txt 2>&1 || tail -5 $D/stdout.txt
  E=$(date +%s.%N)
  echo "OIP_TIFF_GPU_LZW=$m wall $(python3 -c "print(round($E-$S,3))") s, product $(stat -c %s $D/out.TIFF) bytes, md5 $(md5sum < $D/out.TIFF | cut -c1-12)"
  grep -h "TIMING" $D/stdout.txt | tail -2
done
rm -rf $D
