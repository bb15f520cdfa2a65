// param_getter.hpp -- ROS-free loader of the reference's launch-XML parameter block.
//
// Mirrors autorally_control::loadParams(std::map<std::string,XmlRpc::XmlRpcValue>*, const
// std::string&) (src/path_integral/param_getter.cpp:75-148): the <param name= type= value=>
// children of the FIRST <node> under <launch>, one "$(env X)" expansion per value, types
// str/int/double/bool, first definition of a key wins.  ParamValue plays the role of
// XmlRpc::XmlRpcValue (typed, throws on a mismatching conversion).  Boost.PropertyTree and
// XmlRpc are not available in this image, so the XML subset the launch files use (elements,
// attributes, comments) is parsed directly.
#pragma once

#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace mppi_host {

class ParamValue {
 public:
  enum Type { TypeInvalid = 0, TypeBoolean = 1, TypeInt = 2, TypeDouble = 3, TypeString = 4 };
  ParamValue() {}
  ParamValue(bool v) : type_(TypeBoolean), b_(v) {}
  ParamValue(int v) : type_(TypeInt), i_(v) {}
  ParamValue(double v) : type_(TypeDouble), d_(v) {}
  ParamValue(const std::string &v) : type_(TypeString), s_(v) {}
  ParamValue(const char *v) : type_(TypeString), s_(v) {}
  Type getType() const { return type_; }
  operator bool() const { need(TypeBoolean); return b_; }
  operator int() const { need(TypeInt); return i_; }
  operator double() const { need(TypeDouble); return d_; }
  operator std::string() const { need(TypeString); return s_; }

 private:
  void need(Type t) const
  {
    if (type_ != t) throw std::runtime_error("ParamValue: type error");  // XmlRpcException analogue
  }
  Type type_ = TypeInvalid;
  bool b_ = false;
  int i_ = 0;
  double d_ = 0.0;
  std::string s_;
};

typedef std::map<std::string, ParamValue> ParamMap;

namespace detail {

struct XmlAttr { std::string name, value; };
struct XmlElem { std::string tag; std::vector<XmlAttr> attrs; bool self_closing = false; bool closing = false; size_t end = 0; };

// Next tag at or after pos (comments, <?..?> and <!..> skipped). Returns false at end of text.
inline bool next_tag(const std::string &x, size_t pos, XmlElem &e)
{
  for (;;) {
    pos = x.find('<', pos);
    if (pos == std::string::npos) return false;
    if (x.compare(pos, 4, "<!--") == 0) {
      const size_t c = x.find("-->", pos + 4);
      if (c == std::string::npos) return false;
      pos = c + 3;
      continue;
    }
    if (x.compare(pos, 2, "<?") == 0 || x.compare(pos, 2, "<!") == 0) {
      const size_t c = x.find('>', pos);
      if (c == std::string::npos) return false;
      pos = c + 1;
      continue;
    }
    break;
  }
  size_t i = pos + 1;
  e = XmlElem();
  if (i < x.size() && x[i] == '/') { e.closing = true; i++; }
  while (i < x.size() && !isspace((unsigned char)x[i]) && x[i] != '>' && x[i] != '/') e.tag += x[i++];
  for (;;) {
    while (i < x.size() && isspace((unsigned char)x[i])) i++;
    if (i >= x.size()) return false;
    if (x[i] == '>') { e.end = i + 1; return true; }
    if (x[i] == '/') { e.self_closing = true; i++; continue; }
    XmlAttr a;
    while (i < x.size() && x[i] != '=' && !isspace((unsigned char)x[i])) a.name += x[i++];
    while (i < x.size() && (isspace((unsigned char)x[i]) || x[i] == '=')) i++;
    if (i >= x.size() || (x[i] != '"' && x[i] != '\'')) throw std::runtime_error("launch xml: bad attribute");
    const char q = x[i++];
    while (i < x.size() && x[i] != q) a.value += x[i++];
    i++;
    e.attrs.push_back(a);
  }
}

}  // namespace detail

// loadParams(params, file_path), param_getter.cpp:75-148
inline void loadParams(ParamMap *params, const std::string &file_path)
{
  std::ifstream in(file_path);
  if (!in) throw std::runtime_error("Could not load roslaunch file containing mppi controller params at path: " + file_path);
  std::stringstream ss;
  ss << in.rdbuf();
  const std::string x = ss.str();
  detail::XmlElem e;
  size_t pos = 0;
  // <launch> ... first <node ...>
  bool in_launch = false, found = false;
  while (detail::next_tag(x, pos, e)) {
    pos = e.end;
    if (!e.closing && e.tag == "launch") in_launch = true;
    else if (in_launch && !e.closing && e.tag == "node") { found = true; break; }
  }
  if (!found || e.self_closing) return;
  // key / value / type persist from one child to the next exactly like the reference's locals
  std::string key, string_val, param_type = "str";
  int depth = 0;
  while (detail::next_tag(x, pos, e)) {
    pos = e.end;
    if (e.closing) {
      if (depth == 0) break;  // </node>
      depth--;
      continue;
    }
    const bool direct_child = (depth == 0);
    if (!e.self_closing) depth++;
    if (!direct_child) continue;
    for (const detail::XmlAttr &a : e.attrs) {
      if (a.name == "name") key = a.value;
      else if (a.name == "type") {
        param_type = a.value;
        if (param_type != "str" && param_type != "int" && param_type != "double" && param_type != "bool")
          param_type = "str";
      } else if (a.name == "value") {
        string_val = a.value;
        const size_t ps = string_val.find("$(env");
        if (ps != std::string::npos) {
          const size_t pe = string_val.find(')');
          const std::string env = string_val.substr(ps + 6, pe - ps - 6);
          const char *ev = std::getenv(env.c_str());
          if (!ev) throw std::runtime_error("launch xml: environment variable '" + env + "' is not set");
          string_val = string_val.substr(0, ps) + ev + string_val.substr(pe + 1);
        }
        if (string_val.find("$(env") != std::string::npos)
          throw std::runtime_error("Not configured for multiple env variables! '" + string_val + "'");
      }
    }
    ParamValue val;
    if (param_type == "int") val = ParamValue(std::stoi(string_val));
    else if (param_type == "double") val = ParamValue(std::stod(string_val));
    else if (param_type == "bool") val = ParamValue(string_val == "true");
    else val = ParamValue(string_val);
    if (params->find(key) == params->end() && !key.empty()) (*params)[key] = val;
  }
}

}  // namespace mppi_host
