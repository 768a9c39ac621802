/* Build glue for oracle/_ref only (used in place of decoder/macroblock.cc for the ENCODER
 * tools): the reference defines Block<16>::inter_predict(mv, SafeRaster, out) only under
 * HAVE_SSE2 (prediction.cc:680-736).  Without an assembler we route it through the
 * reference's own C++ safe_inter_predict() (prediction.cc:919-971), which is bit-identical
 * to the asm path by construction (same libvpx filters).  Only the encoder's motion search
 * calls this overload. */
#include "macroblock.cc"

template <unsigned int size>
void VP8Raster::Block<size>::inter_predict( const MotionVector & mv,
                                            const SafeRaster & reference,
                                            TwoDSubRange<uint8_t, size, size> & output ) const
{
  const int source_column = column_ * size + ( mv.x() >> 3 );
  const int source_row = row_ * size + ( mv.y() >> 3 );
  safe_inter_predict( mv, reference, source_column, source_row, output );
}

template void VP8Raster::Block<16>::inter_predict( const MotionVector &, const SafeRaster &,
                                                   TwoDSubRange<uint8_t, 16, 16> & ) const;
