// compat/nvbio/basic/html.h -- the tiny HTML writer nvBowtie's report generator prints through (nvbio/basic/html.h:36-174, html.cpp):
// document / head / body / table brackets, rows and cells whose attributes come as NULL-terminated (key, value) C-string pairs, and the
// scoped *_object forms.  Header-only; style() is this layer's own compact stylesheet for the class names the report uses.
#pragma once
#include "types.h"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace nvbio {
namespace html {

enum Formatting { FORMATTED };

namespace priv {
/// "<tag k = "v" ...>" from a NULL-terminated list of (key, value) pairs
inline void open_tag(FILE* output, const char* tag, va_list& args)
{
    fprintf(output, "<%s", tag);
    for (const char* key = va_arg(args, const char*); key != NULL; key = va_arg(args, const char*))
    {
        const char* value = va_arg(args, const char*);
        fprintf(output, " %s = \"%s\"", key, value);
    }
    fprintf(output, ">");
}
inline void cell(FILE* output, const char* tag, const char* text, va_list& args) { open_tag(output, tag, args); fprintf(output, "%s</%s>\n", text, tag); }
/// attributes, NULL, then a printf format and its arguments
inline void formatted_cell(FILE* output, const char* tag, va_list& args)
{
    open_tag(output, tag, args);
    const char* format = va_arg(args, const char*);
    vfprintf(output, format, args);
    fprintf(output, "</%s>\n", tag);
}
} // namespace priv

inline void html_begin(FILE* output) { fprintf(output, "<!DOCTYPE HTML PUBLIC \"-//W3C//DTD HTML 4.0 Transitional//EN\">\n<html lang=\"en\">\n"); }
inline void html_end(FILE* output)   { fprintf(output, "</html>\n"); }
/// css: a file name ending in ".css" (linked) or the stylesheet text itself (inlined)
inline void header(FILE* output, const char* title, const char* css, const char* meta = NULL)
{
    fprintf(output, "<head>\n<meta http-equiv=\"Content-Type\" content=\"text/html; charset=ISO-8859-1\">\n");
    if (meta) fprintf(output, "%s", meta);
    fprintf(output, "<title>%s</title>\n", title);
    const size_t n = strlen(css);
    if (n >= 4 && strcmp(css + n - 4, ".css") == 0) fprintf(output, "<link rel=\"stylesheet\" href=\"%s\" type=\"text/css\">\n", css);
    else                                            fprintf(output, "<style>\n%s\n</style>\n", css);
    fprintf(output, "</head>\n ");
}
inline void body_begin(FILE* output) { fprintf(output, "<body>\n"); }
inline void body_end(FILE* output)   { fprintf(output, "</body>\n"); }
inline void table_begin(FILE* output, const char* id, const char* cls, const char* caption) { fprintf(output, "<table id = \"%s\" class = \"%s\">\n<caption>%s</caption>\n", id, cls, caption); }
inline void table_end(FILE* output)  { fprintf(output, "</table>\n"); }
inline void tr_begin(FILE* output, ...) { va_list a; va_start(a, output); priv::open_tag(output, "tr", a); fprintf(output, "\n"); va_end(a); }
inline void tr_end(FILE* output)     { fprintf(output, "</tr>\n"); }
inline void th(FILE* output, const char* name, ...) { va_list a; va_start(a, name); priv::cell(output, "th", name, a); va_end(a); }
inline void td(FILE* output, const char* name, ...) { va_list a; va_start(a, name); priv::cell(output, "td", name, a); va_end(a); }

inline const char* style()
{
    return
        "body { background-color:#252525; }\n"
        "span statnum { float:left; width:60px; }\n"
        "span statbar { background-color:#AADD44; color:#AADD44; border:1px solid #555555; float:left; display:inline-block; margin-left:5px; }\n"
        "table.params, table.stats { font-family:Calibri, \"Courier New\", Arial, sans-serif; width:84%; margin-left:8%; margin-right:8%; border-collapse:collapse; }\n"
        "table.params caption { font-size:1.0em; color:#FFFFFF; background-color:#000000; padding:4px 7px; }\n"
        "table.stats caption { font-size:1.0em; color:#AADD44; background-color:#000000; height:24px; padding:12px 7px 4px 7px; }\n"
        "table.params th { width:50%; font-size:0.9em; text-align:left; background-color:#AAAAAA; color:#FFFFFF; border:1px solid #999999; padding:5px 7px 4px 7px; }\n"
        "table.stats th { width:2%; font-size:0.9em; text-align:left; background-color:#444444; color:#DDDDDD; border:1px solid #383838; padding:5px 7px 4px 7px; }\n"
        "table.params td { font-size:0.8em; border:1px solid #CBCBCB; padding:3px 7px 2px 7px; background-color:#EBEBEB; }\n"
        "table.stats td { font-size:0.8em; border:1px solid #BBBBBB; padding:3px 7px 2px 7px; background-color:#DDDDDD; }\n"
        "table.params tr.alt td { background-color:#DADADA; }\n"
        "table.stats tr.alt td { background-color:#EAEAEA; }\n"
        "table.stats td.small, table.stats td.smallpink { font-size:0.7em; }\n"
        "table.stats td.red { background-color:#FF9988; border:1px solid #BB6655; }\n"
        "table.stats td.green { background-color:#DCFF9A; border:1px solid #AACC89; }\n"
        "table.stats td.pink, table.stats td.smallpink { background-color:#FFE5D5; border:1px solid #BB9988; }\n"
        "table.stats td.azure { background-color:#AAE5FF; }\n"
        "table.stats td.gray { background-color:#C5C5C5; border:1px solid #999999; }\n"
        "table.stats td.orange { background-color:#FFCC66; border:1px solid #DD9911; }\n"
        "table.stats td.yellow { background-color:#FFFF88; border:1px solid #CCCC33; }\n"
        "table.params tr a, table.stats tr a { color:#DDDDFF; }\n";
}

struct html_object   { html_object(FILE* output) : m_output(output) { html_begin(m_output); } ~html_object() { html_end(m_output); } FILE* m_output; };
struct header_object { header_object(FILE* output, const char* title, const char* css, const char* meta = NULL) : m_output(output) { header(m_output, title, css, meta); } FILE* m_output; };
struct body_object   { body_object(FILE* output) : m_output(output) { body_begin(m_output); } ~body_object() { body_end(m_output); } FILE* m_output; };
struct table_object
{
    table_object(FILE* output, const char* id, const char* cls, const char* caption) : m_output(output) { table_begin(m_output, id, cls, caption); }
    ~table_object() { table_end(m_output); }
    FILE* m_output;
};
struct tr_object
{
    tr_object(FILE* output, ...) : m_output(output) { va_list a; va_start(a, output); priv::open_tag(output, "tr", a); fprintf(output, "\n"); va_end(a); }
    ~tr_object() { tr_end(m_output); }
    FILE* m_output;
};
struct th_object
{
    th_object(FILE* output, const char* name, ...) : m_output(output) { va_list a; va_start(a, name); priv::cell(output, "th", name, a); va_end(a); }
    th_object(FILE* output, const Formatting formatted, ...) : m_output(output) { va_list a; va_start(a, formatted); priv::formatted_cell(output, "th", a); va_end(a); }
    FILE* m_output;
};
struct td_object
{
    td_object(FILE* output, const char* name, ...) : m_output(output) { va_list a; va_start(a, name); priv::cell(output, "td", name, a); va_end(a); }
    td_object(FILE* output, const Formatting formatted, ...) : m_output(output) { va_list a; va_start(a, formatted); priv::formatted_cell(output, "td", a); va_end(a); }
    FILE* m_output;
};

} // namespace html
} // namespace nvbio
