// ivf_to_y4m: what the reference's `vp8decode` front-end does (frontend/vp8decode.cc:43-101), on the MI355X decode path
// through the C++ shim: decode an IVF and write the shown frames as YUV4MPEG2.
//
//   ivf_to_y4m [-s decoder.state] [-o out.y4m] input.ivf
//     -s   continue from a decoder state file in the reference's wire format (written by either implementation;
//          the reference's minihash check of state against IVF header is the one thing not done)
//     -o   output file; without it the stream is decoded and discarded (useful for timing)
//     -l   look-ahead in frames (default 16; 0 = frame by frame, the reference's way): that many frames of the file are handed to the
//          library at a time and their entropy decode runs frame-parallel on its host lanes (FilePlayer::set_look_ahead)
//
//   g++ -std=c++14 -O2 -Iinclude examples/ivf_to_y4m.cc -Lalfalfa_amd/lib -lalfalfa_amd -Wl,-rpath,$PWD/alfalfa_amd/lib
#define ALFALFA_AMD_GLOBAL_NAMES
#include "alfalfa_amd/alfalfa.hh"

#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>

namespace {

struct Options
{
  std::string input, output, state;
  unsigned int look_ahead = 16;
  bool ok = false;
};

Options parse_command_line( int argc, char * argv[] )
{
  Options o;
  for ( int i = 1; i < argc; i++ ) {
    const bool takes_value = !std::strcmp( argv[i], "-s" ) || !std::strcmp( argv[i], "-o" ) || !std::strcmp( argv[i], "-l" );
    if ( takes_value ) {
      if ( i + 1 >= argc ) return o;
      if ( argv[i][1] == 'l' ) o.look_ahead = static_cast<unsigned int>( std::strtoul( argv[i + 1], nullptr, 10 ) );
      else ( argv[i][1] == 's' ? o.state : o.output ) = argv[i + 1];
      i++;
    } else if ( argv[i][0] == '-' && argv[i][1] != '\0' ) {
      return o;                         // unknown switch
    } else if ( o.input.empty() ) {
      o.input = argv[i];
    } else {
      return o;                         // more than one input
    }
  }
  o.ok = !o.input.empty();
  return o;
}

Player open_player( const Options & o )
{
  if ( o.state.empty() ) return Player( o.input );
  return EncoderStateDeserializer::build<Player>( o.state, o.input );
}

// header in front of the first frame (taken from that frame's display size: yuv4mpeg.cc:44-50), then FRAME records
size_t play( Player & player, FileDescriptor * sink )
{
  size_t written = 0;
  while ( !player.eof() ) {
    const RasterHandle shown = player.advance();
    if ( !sink ) continue;
    if ( written == 0 ) sink->write( YUV4MPEGHeader( shown ).to_string() );
    YUV4MPEGFrameWriter::write( shown, *sink );
    written++;
  }
  return written;
}

} // namespace

int main( int argc, char * argv[] )
{
  const Options opt = parse_command_line( argc, argv );
  if ( !opt.ok ) {
    std::cerr << "Usage: " << ( argc > 0 ? argv[0] : "ivf_to_y4m" ) << " [-s decoder_state] [-o y4m_output] [-l look_ahead_frames] input_file\n";
    return EXIT_FAILURE;
  }
  try {
    Player player = open_player( opt );
    player.set_look_ahead( opt.look_ahead );
    if ( opt.output.empty() ) {
      play( player, nullptr );
    } else {
      FileDescriptor sink( std::fopen( opt.output.c_str(), "wb" ) );
      play( player, &sink );
    }
  } catch ( const std::exception & e ) {
    print_exception( argv[0], e );
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}
