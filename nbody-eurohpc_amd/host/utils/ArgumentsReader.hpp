// Minimal command-line reader with the behaviour of the reference's Arguments_reader
// (reference src/common/utils/ArgumentsReader.cpp:20-63): an argument named "x" is matched by the
// token "-x" — so the tag "-im" is written "--im" on the command line — and takes the next token as
// its value when it is declared with a value name.  Required arguments must all be present.
#ifndef ARGUMENTS_READER_HPP_
#define ARGUMENTS_READER_HPP_

#include <map>
#include <string>
#include <vector>

class Arguments_reader {
    std::vector<std::string> argv_;
    std::map<std::string, std::string> required_, optional_, found_, doc_;

  public:
    Arguments_reader(int argc, char **argv);
    // value name "" = flag without value.  Returns false when a required argument is missing.
    bool parse_arguments(const std::map<std::string, std::string> &requireArgs,
                         const std::map<std::string, std::string> &facultativeArgs);
    bool exist_argument(const std::string &tag) const { return found_.count(tag) != 0; }
    std::string get_argument(const std::string &tag) const;
    bool parse_doc_args(const std::map<std::string, std::string> &docArgs);
    void print_usage() const;
};

#endif
