// decode_chain: the job of the reference's `xc-decode-bundle` (frontend/decode-bundle.cc:46-108) on the MI355X decode path:
// a video cut into consecutive IVF pieces -- names read from standard input, one per line -- is decoded by ONE decoder
// that is carried from piece to piece, optionally starting from a decoder state file (reference wire format), and
// written to standard output as a single YUV4MPEG2 stream.  Not done: the reference's minihash check of each piece's
// IVF header against the decoder (boost::hash_combine based) and its state dumps on stderr.
//
//   printf 'a.ivf\nb.ivf\n' | decode_chain [start.state] > out.y4m
//   g++ -std=c++14 -O2 -Iinclude examples/decode_chain.cc -Lalfalfa_amd/lib -lalfalfa_amd -Wl,-rpath,$PWD/alfalfa_amd/lib
#define ALFALFA_AMD_GLOBAL_NAMES
#include "alfalfa_amd/alfalfa.hh"

#include <iostream>
#include <memory>
#include <string>
#include <vector>

namespace {

std::vector<std::string> read_piece_names( std::istream & in )
{
  std::vector<std::string> names;
  for ( std::string line; std::getline( in, line ); ) if ( !line.empty() ) names.push_back( line );
  return names;
}

class Chain
{
  std::unique_ptr<FramePlayer> player_;
  FileDescriptor out_ { 1 };            // stdout
  const char * start_state_;

  void start( const IVF & first )
  {
    if ( start_state_ ) {
      player_.reset( new FramePlayer( EncoderStateDeserializer::build<FramePlayer>( start_state_ ) ) );
      if ( first.width() != player_->width() || first.height() != player_->height() ) throw Unsupported( "state vs. file dimension mismatch" );
    } else {
      player_.reset( new FramePlayer( first.width(), first.height() ) );
    }
    out_.write( YUV4MPEGHeader( player_->example_raster() ).to_string() );
  }

public:
  explicit Chain( const char * start_state ) : start_state_( start_state ) {}

  void append( const std::string & name )
  {
    const IVF piece( name );
    std::cerr << name << ": " << piece.frame_count() << " frames, " << piece.width() << "x" << piece.height() << "\n";
    if ( !player_ ) start( piece );
    for ( uint32_t i = 0; i < piece.frame_count(); i++ ) {
      const Optional<RasterHandle> shown = player_->decode( piece.frame( i ) );     // hidden frames only move the references
      if ( shown.initialized() ) YUV4MPEGFrameWriter::write( shown.get(), out_ );
    }
  }
};

} // namespace

int main( int argc, char * argv[] )
{
  if ( argc > 2 ) {
    std::cerr << "Usage: " << argv[0] << " [starting_state]   (piece names on standard input)\n";
    return EXIT_FAILURE;
  }
  try {
    Chain chain( argc == 2 ? argv[1] : nullptr );
    for ( const std::string & name : read_piece_names( std::cin ) ) chain.append( name );
  } catch ( const std::exception & e ) {
    print_exception( argv[0], e );
    return EXIT_FAILURE;
  }
  return EXIT_SUCCESS;
}
