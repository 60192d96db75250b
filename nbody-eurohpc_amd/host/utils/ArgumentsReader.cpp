#include "utils/ArgumentsReader.hpp"

#include <iostream>

Arguments_reader::Arguments_reader(int argc, char **argv) : argv_(argv, argv + argc) {}

bool Arguments_reader::parse_arguments(const std::map<std::string, std::string> &requireArgs,
                                       const std::map<std::string, std::string> &facultativeArgs)
{
    required_ = requireArgs;
    optional_ = facultativeArgs;
    found_.clear();
    size_t nRequired = 0;
    for (size_t pos = 0; pos < argv_.size(); ++pos) {
        for (const auto *table : {&required_, &optional_}) {
            for (const auto &[tag, valueName] : *table) {
                if ("-" + tag != argv_[pos]) continue;
                if (valueName.empty()) found_[tag] = "";
                else if (pos + 1 < argv_.size()) found_[tag] = argv_[pos + 1];
                else continue;
                if (table == &required_) ++nRequired;
            }
        }
    }
    return nRequired >= required_.size();
}

std::string Arguments_reader::get_argument(const std::string &tag) const
{
    auto it = found_.find(tag);
    return it == found_.end() ? std::string() : it->second;
}

bool Arguments_reader::parse_doc_args(const std::map<std::string, std::string> &docArgs)
{
    if (docArgs.empty()) return false;
    for (const auto *table : {&required_, &optional_})
        for (const auto &kv : *table)
            if (!docArgs.count(kv.first)) return false;
    doc_ = docArgs;
    return true;
}

void Arguments_reader::print_usage() const
{
    std::cout << "Usage: " << argv_[0];
    for (const auto &[tag, valueName] : required_) std::cout << " -" << tag << " " << valueName;
    for (const auto &[tag, valueName] : optional_)
        std::cout << " [-" << tag << (valueName.empty() ? "" : " " + valueName) << "]";
    std::cout << std::endl << std::endl;
    for (const auto *table : {&required_, &optional_})
        for (const auto &[tag, valueName] : *table) {
            auto d = doc_.find(tag);
            std::cout << "\t-" << tag << "\t\t" << (d == doc_.end() ? "" : d->second) << std::endl;
        }
    std::cout << std::endl;
}
